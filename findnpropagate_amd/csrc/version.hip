// Library identification for the ctypes loader (findnpropagate_amd/lib.py).
#include "common.h"

#define FNP_ABI_VERSION 12   // 12 (round 5): the wide-tile entry points of ABI 10 are gone (measured slower in every form)

extern "C" const char *fnp_version(void) { return "fnp-hip gfx950 abi1"; }
extern "C" int fnp_abi_version(void) { return FNP_ABI_VERSION; }
