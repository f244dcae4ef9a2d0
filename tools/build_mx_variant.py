#!/usr/bin/env python3
"""A second build of the library that differs in gemm_mx.hip only: timing ablations of the MX kernel, results wrong on purpose.
  python tools/build_mx_variant.py libribca_mx_noa.so -DMXDBG_NOA ; then RIBCA_LIB=libribca_mx_noa.so python tools/bench_mx_only.py
Switches (gemm_mx.hip / gemm_epi.h): MXDBG_NOA / MXDBG_NOW (no A / W requests), MXDBG_NOCONV (no hi -> fp6 conversion), MXDBG_NOF16 (no f16
MFMAs), MXDBG_NOZ (no residual K steps), MXDBG_NOEPI (no epilogue), MXDBG_NOGELU, MXDBG_NOEMIT (fc1 epilogue without GELU / without the MX3
emission), MXDBG_PF=<n> (LDS prefetch depth of the f16 phase), MXDBG_STAMP (phase cycle sums behind the statistics: tools/stamp_mx.py),
MXDBG_LO4 (round 5: half the lo bytes requested and read, the A lo x W hi' instruction issued in the fp6 format -- the K loop's share of what an
fp4 / fp6 image of A lo could buy; profiles/r5/ab_lo4_timing.txt).  tools/build_ab_lib.py builds BOTH libraries (product + test hooks) of a variant.
DO NOT build -DMXDBG_NOMX: with the scaled MFMAs gone the compiler treats the registers the inline-asm loads fill as dead and reuses them
while the loads are in flight -- the variant faulted on the GPU (round 4); the switch is only left in the source as a marker."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiplexed_image_annotator_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
B.build(verbose=False)
objdir = os.path.join(B.HERE, "build")
obj = os.path.join(B.HERE, "build_" + os.path.splitext(name)[0] + "_gemm_mx.o")
subprocess.run([B._hipcc()] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, "gemm_mx.hip"), "-o", obj], check=True)
objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemm_mx.hip"] + [obj]
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + B.EXPORTS, "-o", os.path.join(B.HERE, name)] + objs, check=True)
print("built", name)
