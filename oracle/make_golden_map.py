"""Golden vectors for the mAP restatement (ccnet_amd/utils/map.py), produced by the REAL reference `utils.map.get_map`
and `voc_ap` in this container.  Test infrastructure only; nothing here runs on the GPU box.

    python oracle/make_golden_map.py      -> tests/golden/map_case.json

cv2 / coloredlogs / matplotlib / pycocotools are not installed; get_map(draw_plot=False) without an images_optional directory never
touches them, so empty stand-in modules satisfy the module-level imports (same approach as the torchvision stub of
oracle/make_golden.py).
"""
import json
import os
import random
import sys
import tempfile
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "map_case.json")


def _stub():
    for name in ("cv2", "coloredlogs", "matplotlib", "matplotlib.pyplot", "pycocotools", "pycocotools.coco", "pycocotools.cocoeval"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.modules["matplotlib"].use = lambda *a, **k: None
    sys.modules["pycocotools.coco"].COCO = object
    sys.modules["pycocotools.cocoeval"].COCOeval = object


def make_case(seed, n_img=8, n_cls=3):
    rnd = random.Random(seed)
    gt, dets = [], []
    for _ in range(n_img):
        g, d = [], []
        for _ in range(rnd.randint(0, 5)):
            c = rnd.randrange(n_cls)
            l, t = rnd.randint(0, 150), rnd.randint(0, 150)
            w, h = rnd.randint(8, 60), rnd.randint(8, 60)
            g.append([c, l, t, l + w, t + h])
            r = rnd.random()
            if r < 0.8:   # a detection near it, sometimes shifted enough to fail the stricter thresholds
                j = rnd.choice([0, 0, 1, 2, 4, 8])
                d.append([c, round(rnd.uniform(0.05, 0.99), 4), l + rnd.randint(-j, j), t + rnd.randint(-j, j), l + w + rnd.randint(-j, j), t + h + rnd.randint(-j, j)])
                if rnd.random() < 0.3:  # duplicate match -> false positive
                    d.append([c, round(rnd.uniform(0.05, 0.99), 4), l + 1, t, l + w, t + h - 1])
        for _ in range(rnd.randint(0, 3)):  # unrelated boxes
            l, t = rnd.randint(0, 150), rnd.randint(0, 150)
            d.append([rnd.randrange(n_cls), round(rnd.uniform(0.01, 0.9), 4), l, t, l + rnd.randint(5, 50), t + rnd.randint(5, 50)])
        gt.append(g)
        dets.append(d)
    return gt, dets


def run_reference(gt, dets, thresholds):
    from utils.map import get_map
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "ground_truth"))
        os.makedirs(os.path.join(root, "detection_results"))
        for i, (g, d) in enumerate(zip(gt, dets)):
            with open(os.path.join(root, "ground_truth", "img%03d.txt" % i), "w") as f:
                for c, l, t, r, b in g:
                    f.write("cls%d %d %d %d %d\n" % (c, l, t, r, b))
            with open(os.path.join(root, "detection_results", "img%03d.txt" % i), "w") as f:
                for c, conf, l, t, r, b in d:
                    f.write("cls%d %s %d %d %d %d\n" % (c, str(conf)[:6], l, t, r, b))
        out = []
        for thr in thresholds:
            out.append(float(get_map(min_overlap=thr, draw_plot=False, score_threhold=0.5, path=root)))
            # get_map leaves its per-threshold scratch in place; the .temp_files json carries the "used" flags
            import shutil
            shutil.rmtree(os.path.join(root, ".temp_files"), ignore_errors=True)
        return out


def main():
    _stub()
    sys.path.insert(0, REF)
    from utils.map import voc_ap
    thresholds = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]
    cases = []
    for seed in (0, 1, 2):
        gt, dets = make_case(seed)
        cases.append({"seed": seed, "num_classes": 3, "gt": gt, "dets": dets, "thresholds": thresholds, "map": run_reference(gt, dets, thresholds)})
    rnd = random.Random(7)
    ap_cases = []
    for _ in range(6):
        n = rnd.randint(0, 12)
        rec = sorted(round(rnd.random(), 3) for _ in range(n))
        prec = [round(rnd.random(), 3) for _ in range(n)]
        ap_cases.append({"rec": rec, "prec": prec, "ap": float(voc_ap(rec[:], prec[:])[0])})
    with open(OUT, "w") as f:
        json.dump({"get_map": cases, "voc_ap": ap_cases}, f)
    print("wrote", OUT, [c["map"][:3] for c in cases])


if __name__ == "__main__":
    main()
