#!/bin/bash
# Reproducer for the finding in DESIGN.md section 3: rebuild the f16 fused kernel on the double-rate
# v_mfma_f32_32x32x16_f16 (in the code shapes given as arguments) and run the launch-to-launch determinism check.
# usage: tools/repro_mfma_k16.sh ["-DRN_XYZ_GROUP=2" ...]    (each argument = one build; -DRN_MFMA_K16=1 is always added)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig_k16.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_fused_h16.o")
[ $# -eq 0 ] && set -- ""
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DRN_MFMA_K16=1 $v -c rad-nerf_amd/csrc/rn_fused_h16.hip -o /tmp/h16_k16.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/h16_k16.o -o $SO || exit 1
  echo "== f16 kernel on v_mfma_f32_32x32x16_f16, extra flags: '$v' =="
  python tools/check_determinism.py --launches 16 | grep f16
done
cp /tmp/orig_k16.so $SO
echo "== tree build (v_mfma_f32_32x32x8f16) =="
python tools/check_determinism.py --launches 16 | grep f16
