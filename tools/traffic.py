"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/traffic.json.

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB
(x 1024), and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (16 B per lane), so the read side is doubled; WRITE_SIZE is exact for
16-B-per-lane streaming stores (other widths uncalibrated: stated in the output).

usage: python tools/traffic.py <fetch_counter_csv> <write_counter_csv> <entities> <meshlets_per_entity> [stream|buffer]
-> gpurun_out/traffic.json (buffer) or gpurun_out/traffic_stream.json (stream: the derived meshlet streams)
"""
import collections
import csv
import json
import os
import re
import sys


def mean_per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, entities, mpe = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    source = sys.argv[5] if len(sys.argv) > 5 else "buffer"
    fetch, launches = mean_per_kernel(fetch_csv, "FETCH_SIZE")
    write, _ = mean_per_kernel(write_csv, "WRITE_SIZE")
    import hashlib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "orbit_amd", "liborbit_cull.so"), "rb") as f:
        lib_sha = hashlib.sha256(f.read()).hexdigest()
    # what the counters were measured on: bench.py reports `traffic_current` by comparing this with the library it runs
    out = {"entities": entities, "meshlets_per_entity": mpe, "meshlet_source": source, "library_sha256": lib_sha,
           "git_commit": os.environ.get("ORBIT_GIT_COMMIT"), "kernels": {},
           "corrections": "KiB -> bytes (x1024); FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B; calibrated "
                          "on wide streaming reads only — the x2 of a kernel that gathers, like the stream path's "
                          "meshlet_emit, is an upper bound); WRITE_SIZE as reported"}
    for k in sorted(fetch):
        if "orbit" not in k:
            continue
        m = re.search(r"(\w+)(?:<[^>]*>)?\((?:orbit::|unsigned|float)", k)
        name = m.group(1) if m else k
        # template variants of one kernel (the stream run also times a few frames of the Meshlet-buffer path): the
        # variant with the most launches is the run's own
        if name in out["kernels"] and out["kernels"][name]["launches"] >= launches[k]:
            continue
        rd = fetch[k] * 1024.0 * 2.0
        wr = write.get(k, 0.0) * 1024.0
        out["kernels"][name] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr,
                                "raw_fetch_kib": fetch[k], "raw_write_kib": write.get(k, 0.0),
                                "launches": launches[k], "kernel_name": k[:120]}
    dom = out["kernels"].get("meshlet_eval_kernel")
    out["dominant_kernel"] = "meshlet_eval_kernel"
    out["hbm_bytes_per_launch"] = dom["hbm_bytes"] if dom else None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "gpurun_out", "traffic.json" if source == "buffer" else "traffic_stream.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
