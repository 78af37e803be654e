#!/usr/bin/env python3
"""HBM bytes per position of every kernel from two rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected
separately with --kernel-trace only).  usage: tools/traffic.py FETCH_DIR WRITE_DIR POSITIONS OUT.json
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B); counters
are in KB.  Only the LAST dispatch of every kernel is used (the timed step; earlier ones are warm-up)."""
import csv, glob, json, re, sys, collections

STAGE = [("k_emis_gain_lane", "emission_rows"), ("k_emis_lane", "emission_rows"), ("k_vit_gain_lane", "viterbi_speculate"),
         ("k_vit_lane", "viterbi_speculate"), ("k_vit_stitch", "viterbi_speculate"), ("k_vit_links", "viterbi_speculate"), ("k_vit_runs", "viterbi_speculate"),
         ("k_vit_spec", "viterbi_speculate"), ("k_vit_fix", "viterbi"), ("k_vit_coop", "viterbi"), ("k_tb_", "traceback"),
         ("k_fused_fwd", "forward_pass"), ("k_fb_fix<36, 0", "forward_pass"), ("k_fused_bwd", "backward_posterior_pass"),
         ("k_fb_fix<36, 1", "backward_chain"), ("k_fb_itemlinks", "links"), ("k_fb_stitch", "links"), ("k_fb_runs", "links"),
         ("k_fused_rowindex", "forward_pass"), ("k_repack_obs", "setup_once"), ("k_poison_dead", "backward_chain"),
         ("k_fb_probe", "forward_pass"), ("k_estep_xi", "estep_reduce"), ("k_estep_hist", "estep_reduce"),
         ("k_vit_place", "viterbi_speculate"), ("k_fb_lane", "forward_backward_speculate"), ("k_combine_lane", "posterior_combine")]


def last_dispatch(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        per = collections.defaultdict(list)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void tehmm::", "").replace("tehmm::", "")
            per[k].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in per.items():
            # one dispatch id may appear once per XCD/dimension: sum the entries of the last id
            last = max(i for i, _ in v)
            out[k] = sum(x for i, x in v if i == last)
    return out


def main():
    fdir, wdir, pos, outp = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    fe, wr = last_dispatch(fdir, "FETCH_SIZE"), last_dispatch(wdir, "WRITE_SIZE")
    per_k, stage = {}, collections.defaultdict(float)
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("k_"):
            continue
        f = 2.0 * fe.get(k, 0.0) * 1024.0 / pos
        w = wr.get(k, 0.0) * 1024.0 / pos
        per_k[k] = [round(f, 1), round(w, 1)]
        st = next((s for p, s in STAGE if k.startswith(p)), "other")
        stage[st] += f + w
    total = sum(v for s, v in stage.items() if s != "setup_once")
    stage = {k: round(v, 1) for k, v in stage.items()}
    stage["total"] = round(total, 1)
    json.dump({"note": __doc__, "positions": pos, "per_kernel_fetch_write_bytes_per_position": per_k,
               "hbm_bytes_per_position": stage}, open(outp, "w"), indent=1)
    print(json.dumps(stage))


if __name__ == "__main__":
    main()
