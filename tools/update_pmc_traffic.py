#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel summaries of tools/refresh_profiles.sh's counter passes: for each workload the
k_accum launch class that moves the most bytes (the per-step primary MSM(T) launches), its HBM bytes per launch, the command and the sha of the
kernels' source at the time of the passes (bench.py flags the figure stale when that source changes).
usage: update_pmc_traffic.py <round tag> <summary dir>      (e.g. r03 profiles)"""
import json
import os
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402

CMD = {"ivc": ("contrast_step_HD_ivc", "--steps 96 (IVC mode, three merged segments)"),
       "accumulator": ("contrast_step_HD", "--steps 96 --mode accumulator"),
       "ivc_4K": ("contrast_step_4K_ivc", "--transformation contrast --resolution 4K --steps 48 --warmup 12"),
       "ivc_8K": ("resize_step_8K_ivc", "--transformation resize --resolution 8K --steps 48 --warmup 12")}


def main():
    tag, d = sys.argv[1], sys.argv[2]
    root = bench.ROOT
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for mode, (key, cmd) in CMD.items():
        f = os.path.join(d, f"{tag}_pmc_summary_{mode}.json")
        if not os.path.exists(f):
            continue
        summ = json.load(open(f))
        acc = {k: v for k, v in summ.items() if k.startswith("k_accum") and "grid=" in k}
        if not acc:
            continue
        # the per-step MSM(T) launches: the launch class that moves the most bytes in all (the few larger launches are the merges')
        k = max(acc, key=lambda x: acc[x]["launches"] * acc[x]["hbm_bytes_per_launch"])
        out[key] = {"kernel": k, "hbm_bytes_per_launch": acc[k]["hbm_bytes_per_launch"], "launches": acc[k]["launches"],
                    "method": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --no-cpu-baseline --no-extras --no-compress {cmd}` "
                              f"(tools/refresh_profiles.sh, round {tag}); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md §HBM); "
                              "these are L2 misses, most of them served by the 256 MB Infinity Cache",
                    "source": f"profiles/{tag}_pmc_summary_{mode}.json (separate PMC passes of the same command; not measured inside this run)",
                    "kernel_source_sha": bench._kernel_source_sha()}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: (v["kernel"], round(v["hbm_bytes_per_launch"] / 1e6, 1)) for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
