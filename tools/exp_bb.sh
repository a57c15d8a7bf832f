#!/bin/bash
# timing-only experiments on the backbone kernel (results are wrong by construction); restores the source afterwards
cd /root/repo
cp svdd_amd/csrc/svdd_nets.hip /tmp/nets_orig.hip
run() { make -C svdd_amd/csrc 2>&1 | grep -E " error" ; echo "$1: $(timeout 120 python tools/backbone_microbench.py 256 200 | grep one-launch)"; }
python3 - <<'PY'
p='svdd_amd/csrc/svdd_nets.hip'
s=open(p).read()
a=s.index("template <bool SPT1>\n__global__ __launch_bounds__(512, 2) void backbone_kernel")
k=s[a:]
assert "        V[0] = ap_[0]; V[1] = ap_[1]; }" in k
k=k.replace("        V[0] = ap_[0]; V[1] = ap_[1]; }","        V[0] = float4{bf0[0],bf0[1],bf0[2],bf0[3]}; V[1] = V[0]; (void)ap_; }",1)
s=s[:a]+k
open(p,'w').write(s)
PY
run E1_noAloads
cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip
# E4: no weight loads from L2 (reuse the first tile)
python3 - <<'PY'
p='svdd_amd/csrc/svdd_nets.hip'
s=open(p).read()
s=s.replace("      if (nxt < it_end) {\n        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;","      if (false) {\n        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;")
assert "if (false) {" in s
open(p,'w').write(s)
PY
run E4_noBloads
cp /tmp/nets_orig.hip svdd_amd/csrc/svdd_nets.hip
make -C svdd_amd/csrc 2>&1 | grep -E " error"
