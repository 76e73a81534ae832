timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for v in 1 2 3; do
echo "run $(timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],3), round(d["value"]), round(d["roofline"]["kernel_ms"],4), d["roofline"]["frac"], d["max_translation_error"])')"
done
