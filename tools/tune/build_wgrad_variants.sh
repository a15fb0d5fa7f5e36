# builds tools/tune/bin/libwg_<variant>.so = the library with wg_sage_bwd.hip compiled under one ablation flag each
set -e
cd "$(dirname "$0")/../../cugraph-gnn_amd/csrc"
mkdir -p ../../tools/tune/bin ../../build_tune
OBJS=$(ls ../build/*.o | grep -v wg_sage_bwd.o)
for v in base "$@"; do
  flag=""; [ "$v" != base ] && flag="-DWG_$v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include -I. $flag -c wg_sage_bwd.hip -o ../../build_tune/bwd_$v.o &
done
wait
for v in base "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/tune/bin/libwg_$v.so $OBJS ../../build_tune/bwd_$v.o -ldl
done
ls -la ../../tools/tune/bin/
