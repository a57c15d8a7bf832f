#!/bin/bash
# timing-only experiments on conv_tower_kernel (results wrong by construction)
# The tracked sources are never touched: the kernels are copied to a scratch directory, patched and built THERE, and the
# microbenchmark loads that build through SVDD_HIP_LIB (svdd_amd/_lib.py). The scratch directory is removed on any exit.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
WORK=$(mktemp -d /tmp/svdd_exp.XXXXXX)
trap 'rm -rf "$WORK"' EXIT
fresh() { cp svdd_amd/csrc/*.hip svdd_amd/csrc/Makefile "$WORK"/; }
build() { make -C "$WORK" -j3 INC="$ROOT/include" 2>&1 | grep -E " error"; }
run() { build; echo "$1: $(SVDD_HIP_LIB=$WORK/libsvdd_hip.so timeout 120 python tools/tower_microbench.py 2>&1 | grep 'n=2560 L=200' | cut -c1-60)"; }
edit() { python3 - "$WORK" "$@" <<'PY'
import sys
p=sys.argv[1]+'/svdd_nets.hip'
s=open(p).read()
a=s.index("template <bool SPT1>\n__global__ __launch_bounds__(512, 4) void conv_tower_kernel"); b=s.index("// --------------------------------------------------------- fused dilated-CNN backbone")
k=s[a:b]
for e in sys.argv[2:]:
    if e=="noA":
        assert "          V[0] = ap_[0]; V[1] = ap_[1];  " in k or "V[0] = ap_[0]; V[1] = ap_[1];" in k
        k=k.replace("V[0] = ap_[0]; V[1] = ap_[1];","V[0] = float4{bf[0],bf[1],bf[2],bf[3]}; V[1] = V[0]; (void)ap_;")
    if e=="noB":
        assert "      if (it + 1 < nit) {                                // the next tile's slice flies under the MFMAs" in k
        k=k.replace("      if (it + 1 < nit) {                                // the next tile's slice flies under the MFMAs","      if (it + 1 < nit && a.n == 12345) {")
    if e=="noepi":
        assert "        const float v = acc[r][e] + (res ? act[o] : 0.0f);" in k
        k=k.replace("#pragma unroll\n    for (int r = 0; r < 7; ++r) {\n      if (r == 6 && rh == 1) continue;\n#pragma unroll\n      for (int e = 0; e < 4; ++e) {                      // C/D layout","#pragma unroll\n    for (int r = 0; r < 7 && a.n == 12345; ++r) {\n      if (r == 6 && rh == 1) continue;\n#pragma unroll\n      for (int e = 0; e < 4; ++e) {                      // C/D layout")
        assert "a.n == 12345; ++r" in k
    if e=="noout":
        assert "  for (int e = tid; e < tile_rows * 16; e += 512) {      // final image -> HBM" in k
        k=k.replace("  for (int e = tid; e < tile_rows * 16; e += 512) {      // final image -> HBM","  for (int e = tid; e < tile_rows * 16 && a.n == 12345; e += 512) {   //")
s=s[:a]+k+s[b:]
open(p,'w').write(s)
PY
}
fresh
run baseline
for v in noA noB noepi noout; do edit $v; run $v; fresh; done
edit noA noB noepi noout; run all4; fresh
true
