#!/bin/bash
# One decisive experiment on the K = 16 f16 MFMA finding: does keeping the issuing wave idle for 80
# cycles after every v_mfma_f32_32x32x16_f16 (the instruction and 5 x s_nop 15 in one asm statement: no write to its A / B
# registers can issue meanwhile) remove the launch-to-launch deviations?  The three variant libraries are prebuilt in the build
# container (rad-nerf_amd/lib/variants/, see the build lines in DESIGN.md); this script only swaps them in and runs the
# determinism check ONCE per variant.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig_lib.so
for v in k16 k16_wait k16_wait_group1 k16_wait_lead k8_wait_lead; do
  cp rad-nerf_amd/lib/variants/libradnerf_hip_$v.so $SO
  echo "== variant $v =="
  timeout -k 10 120 python tools/check_determinism.py --launches 16 2>/dev/null | grep "f16 "
done
cp /tmp/orig_lib.so $SO
echo "== tree build (v_mfma_f32_32x32x8f16) =="
timeout -k 10 120 python tools/check_determinism.py --launches 16 2>/dev/null | grep "f16 "
