set -o pipefail
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_fused.o")
for v in 1 2 3 4 6 12; do
  /opt/rocm/bin/hipcc $FLAGS -DRN_TILE_CHUNK=$v -c rad-nerf_amd/csrc/rn_fused.hip -o /tmp/rn_fused.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/rn_fused.o -o $SO || exit 1
  python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('chunk $v fps', round(d['value'],1))"
done
cp /tmp/orig.so $SO
