# Same-box A/B of two builds of the library differing by compile-time switches:
#   bash tools/ab_lib.sh "<flags A>" "<flags B>" <command using $S2F_LIB ...>
# builds spike2former_amd/libs2f_A.so / libs2f_B.so and runs the command alternately (A B A B) with S2F_LIB pointing at each.
cd $GRAFT_REPO_ROOT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
FA="$1"; FB="$2"; shift 2
for V in A B; do
  F=$FA; [ $V = B ] && F=$FB
  make -s -C spike2former_amd/csrc clean > /dev/null; make -s -j8 -C spike2former_amd/csrc FLAGS="$BASE $F" > /dev/null 2>&1
  cp spike2former_amd/libs2f_hip.so spike2former_amd/libs2f_$V.so
done
for V in A B A B; do
  F=$FA; [ $V = B ] && F=$FB
  echo "== [$V: $F]"; S2F_LIB=$GRAFT_REPO_ROOT/spike2former_amd/libs2f_$V.so "$@"
done
