"""Box-side helper: reduce a rocprofv3 counter_collection.csv to per-kernel means for mk:: kernels
(counter means plus the mean dispatch duration from the start/end timestamps)."""
import collections, csv, glob, json, sys
out = {}
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mk::" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
        out[k]["dispatches"] = len(next(iter(v.values())))
        out[k]["duration_ms_mean"] = sum(dur[k].values()) / len(dur[k])
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
