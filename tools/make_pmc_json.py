#!/usr/bin/env python3
"""profiles/rNN_pmc.json (what bench.py's roofline.traffic reads) from the two PMC summaries of tools/gpu_round_end.sh.
usage: make_pmc_json.py <prof_fetch.txt> <prof_write.txt> <cfg> <out.json>"""
import json
import re
import sys


def table(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+\d+\s*$", line.rstrip())
        if m:
            name = m.group(1).strip()
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"^ovg::(feat|gram|chol)::", "", name)
            name = re.sub(r"^ovg::", "", name)
            name = name.split("(")[0]
            out[name] = float(m.group(4))
    return out


def main():
    fetch, write, cfg, dst = table(sys.argv[1]), table(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    kernels = {k: {"FETCH_SIZE_KiB": fetch.get(k, 0.0), "WRITE_SIZE_KiB": write.get(k, 0.0)} for k in sorted(set(fetch) | set(write))}
    doc = {"cfg": cfg,
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2` "
                     "(tools/gpu_round_end.sh); text summaries committed next to this file",
           "note": "KiB per launch; FETCH_SIZE on gfx950 counts half of a wide coalesced read (MI355X_MICROARCH.md, HBM section): bench.py doubles the read side",
           "kernels": kernels}
    json.dump(doc, open(dst, "w"), indent=1)
    tot = sum(1024 * (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) for v in kernels.values())
    print("%d kernels, %.1f MB per update (read side doubled)" % (len(kernels), tot / 1e6))


if __name__ == "__main__":
    main()
