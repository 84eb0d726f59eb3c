import json, sys
d = json.load(open(sys.argv[1]))
for r in d["shapes"]:
    t = sorted(r["tflops"].items(), key=lambda kv: -kv[1])[:6]
    print("%-24s K=%5d N=%4d auto %5.0f | " % (r["tag"], r["K"], r["N"], r["tflops"].get("auto", 0)) + "  ".join("%s %.0f" % (k.replace("_w2x2","").replace("_w4x2","w42").replace("_w2x4","w24"), v) for k, v in t))
