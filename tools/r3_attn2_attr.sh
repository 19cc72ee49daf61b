#!/bin/bash
# attn2_kernel cost attribution: builds libtld_hip_a2d{1,2,3,5}.so (tld_attn.hip with -DTLD_A2_DBG=n) HERE when called with "build";
# on the GPU box runs tools/attn_bench.py against each.
cd "$(dirname "$0")/.."
P=transformer_latent_diffusion_amd
if [ "$1" = build ]; then
  make -C $P/csrc -j >/dev/null || exit 1
  for n in 0 1 2 3 5; do
    /opt/rocm/bin/hipcc -DTLD_A2_DBG=$n -DTLD_A2_CLK=1 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-result -Wno-unused-value \
      -c $P/csrc/tld_attn.hip -o /tmp/build/attn_d$n.o || exit 1
    objs=$(ls $P/csrc/_build/*.o | grep -v tld_attn.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libtld_hip_a2d$n.so $objs /tmp/build/attn_d$n.o || exit 1
  done
  exit 0
fi
mkdir -p gpurun_out/attn2
{
for rep in 1 2; do
  for lib in libtld_hip.so libtld_hip_a2d0.so libtld_hip_a2d1.so libtld_hip_a2d2.so libtld_hip_a2d3.so libtld_hip_a2d5.so; do
    for shape in "1024 32" "4096 8"; do
      set -- $shape
      echo -n "$lib: "; A2_CLK=$( [ $lib = libtld_hip.so ] || echo 1 ) TLD_LIB=$PWD/$P/$lib python tools/attn_bench.py --ntok $1 --batch $2 2>/dev/null | cut -c1-100
    done
  done
done
} 2>&1 | tee gpurun_out/attn2/attr.txt
