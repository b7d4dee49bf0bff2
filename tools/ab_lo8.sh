#!/bin/bash
# re-price byte removal (a 3-byte activation format: lo plane at 1 byte per element) on the kernel classes that are
# HBM-bound since tile 14 -- with TIMING-ONLY builds (-DMPX_ABL_LO8=<mask>, csrc/mpx_conv.h: the lo-plane traffic of a kernel class
# moves half its bytes, same instruction count, WRONG results), never the product library: tools/ab_variants.sh binds them per process.
# One gpurun call, two interleaved passes, batch 2340.
# NEEDS THE TREE OF COMMIT 3102d34: the MPX_ABL_LO8 switches lived in the kernels of that commit (csrc/mpx_conv.h lists them) and were removed again once the
# numbers were in (profiles/r05_lo8_byte_pricing.txt, profiles/r05_convw_instruction_stream_and_btail_dead_columns.txt): product kernels carry no probe code.
#   tools/ab_lo8.sh          product | convw (1) | conv256p (2) | btail (4) | convx (8) | all four (15): 8-B accesses (half lines)
#   tools/ab_lo8.sh lines    product | convw (16) | btail (32) | convx (64) | those three + conv256p (114): every second lo access
#                            out of range, the others whole 16-B accesses (what a byte plane with 16 channels per lane would move)
cd "$(dirname "$0")/.."
export PAT='256->1024|1024->256|128->512|64->64    k3|512->128|conv total'
if [ "$1" = lines ]; then
  exec tools/ab_variants.sh "-DMPX_ABL_LO8=16" "-DMPX_ABL_LO8=32" "-DMPX_ABL_LO8=64" "-DMPX_ABL_LO8=114"
fi
exec tools/ab_variants.sh "-DMPX_ABL_LO8=1" "-DMPX_ABL_LO8=2" "-DMPX_ABL_LO8=4" "-DMPX_ABL_LO8=8" "-DMPX_ABL_LO8=15"
