for ov in 0 1 2; do
  timeout 200 python bench.py --steps 40 --warmup 8 --overlap $ov --no_cpu_baseline --no_roofline 2>/dev/null > /tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('overlap', $ov, round(d['value'],2), round(d['ms_per_step'],3), round(d['host_issue_ms_per_step'],3))"
done
