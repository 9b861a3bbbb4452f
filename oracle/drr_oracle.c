/*
 * drr_oracle.c -- CPU oracle for the DRR rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker that the HIP kernels in
 * diffdrr_amd/csrc are compared against; it is never the thing measured or
 * shipped.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or call it.  Nothing under diffdrr_amd/ imports it, and
 * the product path raises if the HIP library is missing.
 *
 * What it restates (reference = eigenvivek/DiffDRR v0.6.0):
 *   Siddon.forward            diffdrr/renderers.py:34-91
 *   _get_alphas               diffdrr/renderers.py:94-113
 *   _get_alpha_minmax         diffdrr/renderers.py:124-140
 *   _get_xyzs                 diffdrr/renderers.py:143-153
 *   _get_voxel                diffdrr/renderers.py:156-169  (+ aten grid_sampler_3d,
 *                             nearest / bilinear, padding zeros -- third party, PyTorch,
 *                             pinned `pytorch>=2.2` in the reference's environment.yml:8)
 *   reduce                    diffdrr/renderers.py:175-183
 *   Trilinear.forward         diffdrr/renderers.py:205-254
 * plus the analytic derivatives of exactly that program (what torch autograd
 * returns for it), used to check the HIP backward kernels.
 *
 * Pinning: the reference ships no golden vectors for this path ("parity
 * unpinned" by its own tests).  The oracle is therefore pinned
 * against outputs of the unmodified reference executed in the build
 * container: tests/golden/make_golden.py imports /root/reference/diffdrr
 * (through the test-only shims in oracle/ref_shims) and stores inputs, fp32
 * and fp64 outputs and autograd gradients in tests/golden/ (npz files);
 * tests/test_oracle_golden.py checks this file against them.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdlib.h>

#define REAL float
#define SFX f32
#include "drr_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL double
#define SFX f64
#include "drr_oracle_impl.h"
#undef REAL
#undef SFX

int oracle_abi_version(void) { return 1; }
