cd $GRAFT_REPO_ROOT
for D in "" "-DNO_ATOMIC" "-DNO_STAGE" "-DNO_MAC" "-DNO_REDUCE" "-DNO_STAGE -DNO_MAC" "-DNO_STAGE -DNO_MAC -DNO_REDUCE"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics $D -I spike2former_amd/csrc -I include tools/micro/dw_wgrad_probe.hip -o /tmp/dwp 2>/dev/null && echo "[$D]" && /tmp/dwp
done
