"""Shared driver for the reference's OptTest scenes (tests/golden/opt_scenes.json).

A scene is a little program over named vertex sets; ``solve_fn(node_pos, node_dq, node_w, k,
canon, live) -> node_dq_out`` is the implementation under test (CPU oracle or the HIP path),
``warp_fn(node_pos, node_dq, node_w, k, verts) -> warped`` likewise.
"""
import json
import os

import numpy as np

SCN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opt_scenes.json")))


def scene_ids():
    return [s["name"] for s in SCN["scenes"]]


def run_scene(scene, solve_fn, warp_fn):
    """Returns max |warp(src) - expected| over all assert_warp ops, and per-op details."""
    g1, g2 = SCN["nodes"]["g1"], SCN["nodes"]["g2"]
    node_pos = np.array(g1 if scene["nodes"] == "g1" else g1 + g2, np.float32)
    D = len(node_pos)
    node_dq = np.zeros((D, 8), np.float32)
    node_dq[:, 0] = 1.0  # DualQuaternion(0,0,0,0,0,0), opt_optimisation_test.cpp:51
    node_w = np.full(D, SCN["dg_w"], np.float32)
    k = SCN["knn"]
    sets = {n: np.array(v, np.float32) for n, v in scene["sets"].items()}
    worst, log = 0.0, []
    for op in scene["ops"]:
        if op[0] == "solve":
            node_dq = solve_fn(node_pos, node_dq, node_w, k, sets[op[1]], sets[op[2]])
        elif op[0] == "warp":
            sets[op[2]] = warp_fn(node_pos, node_dq, node_w, k, sets[op[1]])
        elif op[0] == "assert_warp":
            got = warp_fn(node_pos, node_dq, node_w, k, sets[op[1]])
            err = float(np.max(np.abs(got - sets[op[2]])))
            log.append((op[1], op[2], err))
            worst = max(worst, err)
    return worst, log
