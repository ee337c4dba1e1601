#!/bin/bash
# Generic compile-flag sweep for one fused kernel file.  usage: FILE=rn_fused MLP=f32 bash tools/gpu_variants_flags.sh "-DA=1 -DB=2" "-DA=2" ...
set -o pipefail
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "$FILE.o")
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $v -c rad-nerf_amd/csrc/$FILE.hip -o /tmp/$FILE.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/$FILE.o -o $SO || exit 1
  echo "== $v"
  for g in ${GRIDSEL:-hash19 tiled16}; do
    timeout -k 10 120 python tools/bench_fused.py --mlp $MLP --grid $g --sweep 8192,206016,1048576 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', d['mlp'], d['grid'], d['M8192_ms'], d['M206016_ms'], d['M1048576_ms'])"
    python bench.py --mlp $MLP --grid $g --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('   bench $MLP $g fps', round(d['value'],1))"
  done
done
cp /tmp/orig.so $SO
