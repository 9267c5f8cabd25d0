cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spike2former_amd/csrc -I include tools/micro/gemm_fwd_probe.hip -o /tmp/gp 2>/dev/null
echo "[2x2 wave tiles]"; /tmp/gp
echo "[2x4 wave tiles, one wave across N]"; S2F_GEMM_WN1=1 /tmp/gp
for D in $GP_VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $D -I spike2former_amd/csrc -I include tools/micro/gemm_fwd_probe.hip -o /tmp/gp 2>/dev/null && echo "[$D]" && /tmp/gp && echo "[$D, WN1]" && S2F_GEMM_WN1=1 /tmp/gp
done
