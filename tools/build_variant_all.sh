# dev tool: a variant of librmcv_hip.so with EVERY unit recompiled under extra flags:  bash tools/build_variant_all.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../rmcv_amd/csrc"
name=$1; flags=$2
mkdir -p /tmp/varall_$name
objs=""
for u in k_binary k_contours k_contours_w4 k_detect k_classify k_pnp rmcv_host rmcv_track rmcv_gather; do
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-value -c $u.hip -o /tmp/varall_$name/$u.o &
  objs="$objs /tmp/varall_$name/$u.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/var_$name.so $objs ../lib/obj/synth.o -ldl
echo built rmcv_amd/lib/var_$name.so
