# A/B of a compile-time switch on ONE GPU box: bash tools/ab_build.sh "-DSWITCH" [bench args]   (rebuilds the library in place, twice)
cd $GRAFT_REPO_ROOT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
for V in "" "$1" "" "$1"; do
  make -s -C spike2former_amd/csrc clean > /dev/null; make -s -j8 -C spike2former_amd/csrc FLAGS="$BASE $V" > /dev/null 2>&1
  echo "[flags: $V] $(python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
make -s -C spike2former_amd/csrc clean > /dev/null; make -s -j8 -C spike2former_amd/csrc > /dev/null 2>&1
