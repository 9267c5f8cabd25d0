cd $GRAFT_REPO_ROOT
for D in ${GP_VARIANTS:-"" "-DGP_NO_SPLIT" "-DGP_NO_STORE" "-DGP_NO_MFMA" "-DGP_NO_LDSREAD" "-DGP_NO_GLOBAL"}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics $D -I spike2former_amd/csrc -I include tools/micro/gemm_dw_probe.hip -o /tmp/gdp 2>/dev/null && echo "[$D]" && /tmp/gdp
done
