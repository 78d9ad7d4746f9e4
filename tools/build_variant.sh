# dev tool: a variant of librmcv_hip.so with one unit recompiled under extra flags
#   bash tools/build_variant.sh <name> <unit.hip> "<flags>"   ->  rmcv_amd/lib/var_<name>.so   (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/../rmcv_amd/csrc"
name=$1; unit=$2; flags=$3
mkdir -p /tmp/var_$name
/opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-value -c $unit -o /tmp/var_$name/${unit%.hip}.o
objs=""
for o in ../lib/obj/*.o; do b=$(basename $o); if [ "$b" = "${unit%.hip}.o" ]; then objs="$objs /tmp/var_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/var_$name.so $objs -ldl
echo built rmcv_amd/lib/var_$name.so
