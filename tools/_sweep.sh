TDK_DVO_VARIANT=9 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for v in 9 9 9; do
echo "VARIANT=$v $(TDK_DVO_VARIANT=$v timeout 100 python bench.py 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],3), round(d["roofline"]["kernel_ms"],4), d["roofline"]["launches"], d["max_translation_error"])')"
done
