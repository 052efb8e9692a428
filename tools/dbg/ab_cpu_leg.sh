for rep in 1 2; do for cs in 0 15; do
python3 bench.py --steps 20 --warmup 5 --cpu-seconds $cs --no-probe 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; o=r["other_configs"]
print("cpu-seconds %s: value %.1f M frac %.4f kernel %.4f replicas %.3f post %.3f | cfg5 %.3f B16384 %.3f" % (sys.argv[1], d["value"]/1e6, r["frac"], r["avg_kernel_ms"], r["replicas"]["frac_median"], r["post_reset_frac"], o["cfg5"][1], o["cfg2-B16384"][1]))' $cs
done; done
