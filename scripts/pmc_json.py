"""rocprofv3 --pmc counter CSVs of `scripts/klt_only.py N R` -> the k_lk summary bench.py reads for roofline.traffic and
valu_util.   python scripts/pmc_json.py <keypoints per launch> <launches> <counter_collection.csv>... > profiles/roundN_k_lk_pmc.json"""
import collections, csv, hashlib, json, os, sys
n_kp, launches = int(sys.argv[1]), int(sys.argv[2])
tot = collections.defaultdict(float)
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        if "k_lk" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256()
for f in ("vo_klt.hip", "vo_image.hip", "vo_common.h"):
    sha.update(open(os.path.join(root, "visual_odometry_amd", "csrc", f), "rb").read())
waves = tot.get("SQ_WAVES", n_kp * launches)
per_wave = {k: v / waves for k, v in tot.items() if k.startswith("SQ_") and k != "SQ_WAVES"}
fetch = tot.get("FETCH_SIZE", 0.0) * 1024 / launches
write = tot.get("WRITE_SIZE", 0.0) * 1024 / launches
cycles = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8 / launches            # the counter sums the 8 XCDs
out = {
    "kernel": "k_lk<Cfg<5,6,9,7,52>, W41> (41x41 window)",
    "command": "rocprofv3 --pmc <one block group per pass> -- python3 scripts/klt_only.py %d %d (scripts/profile_round.sh)" % (n_kp, launches - 1),
    "source_sha256": sha.hexdigest(), "keypoints_per_launch": n_kp, "launches": launches,
    "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
    "hbm_bytes_per_launch": 2 * fetch + write,
    "note": "fabric-side request bytes (L2 misses; Infinity-Cache hits are counted); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes "
            "for 16-B/lane loads (all stagings are 16-B LDS-DMA from the padded pyramid now)",
    "TCC_hit_rate": tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"], 1.0) if "TCC_HIT_sum" in tot else None,
    "per_wave": per_wave,
    "kernel_cycles_per_launch": cycles,
    # VALU issue utilisation against the guide's rate: one wave64 VALU instruction per 2 cycles per SIMD (MI355X_MICROARCH.md),
    # 1024 SIMDs.  By the same formula a kernel of nothing but v_fma_f32 reaches 0.67 at 4 waves per SIMD and 0.75-0.82 at 8
    # (profiles/round2_valu_clock_pmc.txt): that, not 1.0, is what this figure can be compared with
    "valu_issue_util": (tot.get("SQ_INSTS_VALU", 0.0) / launches) * 2.0 / (1024.0 * cycles) if cycles else None,
    "wave_time_share": {k: per_wave[k] / per_wave["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")
                        if k in per_wave and "SQ_WAVE_CYCLES" in per_wave},
}
print(json.dumps(out, indent=1))
