"""Development aid: prints the headline fields of the last JSON line of a bench log."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
s = d['stage_ms_per_step']
print(f"{sys.argv[1]:40s} value={d['value']:8.1f} trace={s['trace']:.3f} ({d['trace_kernel_mrays_per_s']:.0f} Mr/s) shade={s['shade']:.3f} step={d['ms_per_step']:.3f}")
