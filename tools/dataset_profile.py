"""The rows of profiles/rNN_dataset.md: tools/bench_dataset.run in the driver's default configuration (pairs as single library
calls on two worker contexts, overlapping the loading / description of further fragments) and in its plainest one (part by part,
pairs composed from the staged entries in Python), one scene of 60 fragments and the 3DMatch-shaped set, YOHO-O and YOHO-C.

    python tools/dataset_profile.py [out.md]          (about 40 s on one MI355X)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_dataset

CASES = [
    ("one scene: 60 fragments x 5000 kp (2.30 GB), 495 pairs; YOHO-O, PartII for the voted matches", dict(estimator="yohoo", runs=3)),
    ("the same; part by part, pairs composed in Python (round 3's first driver)", dict(estimator="yohoo", runs=2, fused=False, overlap=False)),
    ("the same; YOHO-O, PartII for every match", dict(estimator="yohoo", runs=2, hypotheses="all")),
    ("the same; YOHO-C, 1000 iterations sampled on the device", dict(estimator="yohoc", runs=2)),
    ("3DMatch-shaped: 8 scenes, 433 fragments x 5000 kp (16.63 GB), 1623 pairs; YOHO-O, PartII for the voted matches", dict(estimator="yohoo", runs=2, preset="3dmatch")),
    ("the same 3DMatch-shaped set; part by part, pairs composed in Python", dict(estimator="yohoo", runs=2, preset="3dmatch", fused=False, overlap=False)),
    ("the same 3DMatch-shaped set; YOHO-C, 1000 iterations sampled on the device", dict(estimator="yohoc", runs=2, preset="3dmatch")),
]
lines = ["| test set, estimator, configuration | page cache | total s | keypoints/s end to end | pairs/s end to end | setup s (load + H2D + PartI; pairs run meanwhile) | "
         "disk read s, loader threads (rate) | device waiting for the loader s | setup + pairs wall s | pairs left behind the last description s | "
         "gather + archives + pre.log + RR s | RR |", "|" + "---|" * 12]
raw = []
for label, kw in CASES:
    d = bench_dataset.run(**kw)
    raw.append(d)
    for r in d["runs"]:
        z = r["rank0"]
        lines.append("| %s | %s | %.3f | %d | %.1f | %.3f | %.3f (%.2f GB/s) | %.3f | %.3f | %.3f | %.3f | %.4f |" % (
            label, "dropped" if r["page_cache"].startswith("dropped") else "warm", r["total_s"], r["keypoints_per_s_end_to_end"],
            r["pairs_per_s_end_to_end"], z["setup_s (load + H2D + PartI, overlapped)"], z["disk_read_s (loader thread)"], z["disk_GBps"],
            z["device_waiting_for_loader_s"], z["setup_and_pairs_wall_s"], z["pairs_behind_last_setup_s"], z["gather_write_RR_s"],
            r["registration_recall"]))
text = "\n".join(lines) + "\n\nRaw JSON, one line per configuration:\n\n```\n" + "\n".join(json.dumps(d) for d in raw) + "\n```\n"
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    open(sys.argv[1], "w").write(text)
print(text)
