#!/bin/bash
# Development: build libfnp_hip.so of a git revision (default HEAD) into findnpropagate_amd/csrc/ab/libfnp_abhead.so
# for same-box A/B timing with FNP_LIB_PATH (box-to-box clock differences exceed 10 %).
set -e
REV=${1:-HEAD}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
git -C "$ROOT" archive "$REV" findnpropagate_amd/csrc include | tar -x -C "$T"
make -s -C "$T/findnpropagate_amd/csrc" -j4 >/dev/null
mkdir -p "$ROOT/findnpropagate_amd/csrc/ab"
cp "$T/findnpropagate_amd/libfnp_hip.so" "$ROOT/findnpropagate_amd/csrc/ab/libfnp_abhead.so"
rm -rf "$T"
