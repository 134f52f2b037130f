#!/bin/bash
# NOTE (round 6): -DNRC_DIAG_LASTDIR / -DNRC_DIAG_BISECT left the product source; this tool builds them from the tree of commit aa01da1 (round 5): git worktree add /tmp/r05 aa01da1
# (the product is compiled with -fno-slp-vectorize since the cause was found: these diagnostic builds switch the vectoriser back ON)
# builds the probe library + harness with -DNRC_DIAG_BISECT=<mask> (see new_ray_dir in nrc_integrator.hip): tools/bisect_build.sh <mask>...
cd "$(dirname "$0")/.."
for m in "$@"; do
    make -C nrc-hpm-renderer_amd/csrc ARCH=gfx950 OUT=../lib_bs$m "EXTRA=-fslp-vectorize -DNRC_DIAG_LASTDIR -DNRC_DIAG_LOWPRIO=8 -DNRC_GEN_WAVES_PER_SIMD=4 -DNRC_DIAG_BISECT=$m" > /dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -Iinclude tests/cpp/stress_main.cpp -o tests/cpp/_build/stress_main_bs$m \
        -Lnrc-hpm-renderer_amd/lib_bs$m -lnrc_hpm -pthread "-Wl,-rpath,\$ORIGIN/../../../nrc-hpm-renderer_amd/lib_bs$m" || exit 1
    echo built bs$m
done
