// Library identification for the ctypes loader (findnpropagate_amd/lib.py).
#include "common.h"

#define FNP_ABI_VERSION 14   // 14 (round 6): fnp_gather_counts_host (the counts stored into pinned host memory by the launch itself); 13 (round 6): fnp_rankgrid.counters + fnp_rankgrid_counter_words (counted marks: the rank prefix is one launch); 12 (round 5): the wide-tile entry points of ABI 10 are gone

extern "C" const char *fnp_version(void) { return "fnp-hip gfx950 abi1"; }
extern "C" int fnp_abi_version(void) { return FNP_ABI_VERSION; }
