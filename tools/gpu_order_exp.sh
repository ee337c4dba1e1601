for w in 0 512; do for m in f32 f32x2 f16; do RN_EXP_ORDER_W=$w python bench.py --mlp $m --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('order_w $w bench $m fps', round(d['value'],1))"; done; done
RN_EXP_ORDER_W=512 timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3
