# round 6: the headline's run-time knobs once more on one box (stream groups, first-stage width of the line search)
for a in "--groups 4" "--groups 3" "--groups 2" "--groups 4 --ls-split 3" "--groups 4 --ls-split 5" "--groups 4"; do
  python3 bench.py --steps 20 --warmup 5 --no-unfused --no-cpu-baseline $a > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json;j=json.load(open('/tmp/o.json'));print('$a:', round(j['value'],2), round(j['ms_per_step'],3))"
done
