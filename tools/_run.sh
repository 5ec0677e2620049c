for g in 512 2048 256; do
KGPU_GENERAL_WG=$g python bench.py --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('general wg $g:', d['value']/1e6, d['roofline']['avg_kernel_ms'])"
done
