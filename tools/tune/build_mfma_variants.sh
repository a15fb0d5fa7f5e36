# builds tools/tune/bin/libmf_<variant>.so = the library with wg_sage_mfma.hip compiled under extra flags (tuning: wrong results
# for some): usage: build_mfma_variants.sh name1="-DFLAG ..." name2=...
set -e
cd "$(dirname "$0")/../../cugraph-gnn_amd/csrc"
mkdir -p ../../tools/tune/bin ../../build_tune
OBJS=$(ls ../build/*.o | grep -v wg_sage_mfma.o)
for kv in "$@"; do
  v=${kv%%=*}; flag=${kv#*=}; [ "$flag" = "$kv" ] && flag=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include -I. $flag -c wg_sage_mfma.hip -o ../../build_tune/mf_$v.o &
done
wait
for kv in "$@"; do
  v=${kv%%=*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/tune/bin/libmf_$v.so $OBJS ../../build_tune/mf_$v.o -ldl
done
ls -la ../../tools/tune/bin/
