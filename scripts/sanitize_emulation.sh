#!/bin/bash
# Host-only check (no GPU): the CPU lane emulation of the wave engine - the engine's own source
# with the 64 lanes as a loop - built with AddressSanitizer and UndefinedBehaviorSanitizer and
# run over every iteration mode at both transform sizes (tests/csrc/emu_sanitize.cpp).
# The build takes about two minutes, which is why this is not part of the pytest suite.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/glv_emu_sanitize
g++ -std=c++17 -O1 -g -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=all -static-libasan \
    -I"$ROOT/tts_amd/csrc" -o "$OUT" "$ROOT/tests/csrc/emu_sanitize.cpp"
"$OUT"
