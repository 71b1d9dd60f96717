// tile_shapes.h -- tile widths (column groups of 16 bytes) per pass length, shared by the translation units that
// instantiate tile kernels (each settled by A/B on the GPU: 4 / 16 column groups at 1024, 4 at 2048, ...; profiles/r01_*, r02_*).
#pragma once
#define FOURIER_CG_128_ROWS 32
#define FOURIER_CG_512 8
#define FOURIER_CG_1024 8
#define FOURIER_CG_2048 8
// L = 2048 holds a 256 KiB tile per workgroup at 16 columns -- one workgroup per CU, no overlap of its load and
// compute phases.  Default plans therefore run the FIRST pass on 64-byte-wide tiles (8 columns, 128 KiB, two
// workgroups per CU; the transposed store does not care about the tile width): 6.3-6.6 vs 7.2-7.8 ms per 1024
// transforms of 2^21 (profiles/r02_s2_l2048_and_xcd_fused_ab.jsonl).  FOURIER_WIDE_2048=1 in the environment at plan
// creation brings the 16-column first pass back (A/B).
#define FOURIER_CG_2048_FIRST 4

#define FOURIER_CG_4096 2
