#!/bin/bash
# Development aid: registers, spills, scratch and LDS of the ray kernels of a built csrc directory.
dir=${1:-raytracer_amd/csrc}
tmp=$(mktemp -d)
for o in wavefront traverse; do
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $dir/$o.o /dev/null 2>/dev/null || true
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.co | awk '
/\.name:/ {name=$2}
/\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {vs=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
/\.wavefront_size:/ { if (name ~ /traceKernel/) printf "%-64s vgpr %s spill %s sgpr %s scratch %s lds %s\n", substr(name,1,64), v, vs, s, p, g }'
done
rm -rf $tmp
