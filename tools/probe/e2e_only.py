import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l)
        for k in ("end_to_end_early","end_to_end"):
            if k in d: print(k, d[k]["ms_per_frame"], d[k]["one_frame_lookahead"]["ms_per_frame"], d[k]["serial"]["ms_per_frame"])
