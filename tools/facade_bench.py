#!/usr/bin/env python3
"""Latency of one takeStep through the C++ drop-in classes (tests/cpp/facade_step.cpp) at a synth config (GPU box).
usage: PYTHONPATH=. python tools/facade_bench.py [config=ref] [steps=200] [device_scan=1]
       PYTHONPATH=. python tools/facade_bench.py loop [config=ref] [steps=200] [device_scan=1]
           the node's whole loop (tests/cpp/facade_loop.cpp: takeStep + publishPoseEst through TopDownRenderCore) with the
           range scale — the `res` of render and score — moving on every step like in the node, and with a fixed one"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from top_down_renderer_amd import build, synth  # noqa: E402

PKG = os.path.join(ROOT, "top_down_renderer_amd")


def loop_main(argv):
    name = argv[0] if len(argv) > 0 else "ref"
    steps = int(argv[1]) if len(argv) > 1 else 200
    device_scan = int(argv[2]) if len(argv) > 2 else 1
    build.build()
    exe = os.path.join(tempfile.mkdtemp(prefix="tdr_facade_"), "facade_loop")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_loop.cpp"), "-o", exe, "-L", PKG, "-ltdr_hip",
                    f"-Wl,-rpath,{PKG}"], check=True)
    sc = synth.make_scene(name)
    cfg = sc.cfg
    d = tempfile.mkdtemp(prefix="tdr_facade_run_")
    # the node's defaults: range scale in [0.5, 4], target uncertainty 2.5 m; a known scale (fixed_scale = the map's)
    open(os.path.join(d, "meta.txt"), "w").write(
        f"{cfg.ncls} {cfg.map_size} {cfg.map_size} {cfg.nb} {cfg.nr} {len(sc.pts)} 1 {len(sc.states)} 17 2 "
        f"0.5 4.0 2.5 1.0 {cfg.map_resolution} {device_scan}\n")
    np.ascontiguousarray(np.transpose(sc.class_maps, (0, 2, 1)), np.float32).tofile(os.path.join(d, "maps.bin"))
    np.ascontiguousarray(sc.class_mask.T, np.uint8).tofile(os.path.join(d, "mask.bin"))
    pcl = np.zeros((len(sc.pts), 8), np.float32)
    pcl[:, :3] = sc.pts[:, :3]
    pcl[:, 4] = sc.pts[:, 3]
    pcl.tofile(os.path.join(d, "pts.bin"))
    sc.states.tofile(os.path.join(d, "states.bin"))
    np.tile(np.asarray((1.0, 0.25, 0.01), np.float32), (2, 1)).tofile(os.path.join(d, "motion.bin"))
    env = dict(os.environ, TDR_FACADE_BENCH=str(steps))
    r = subprocess.run([exe, d], capture_output=True, text=True, env=env)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr)
    return r.returncode


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "loop":
        return loop_main(sys.argv[2:])
    name = sys.argv[1] if len(sys.argv) > 1 else "ref"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    device_scan = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    build.build()
    exe = os.path.join(tempfile.mkdtemp(prefix="tdr_facade_"), "facade_step")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_step.cpp"), "-o", exe, "-L", PKG, "-ltdr_hip",
                    f"-Wl,-rpath,{PKG}"], check=True)
    sc = synth.make_scene(name)
    cfg = sc.cfg
    d = tempfile.mkdtemp(prefix="tdr_facade_run_")
    open(os.path.join(d, "meta.txt"), "w").write(
        f"{cfg.ncls} {cfg.map_size} {cfg.map_size} {cfg.nb} {cfg.nr} {len(sc.pts)} {len(sc.states)} {cfg.res} "
        f"{float(cfg.ang_res)!r} 17 1.0 0.25 0.01 {device_scan} 0\n")
    np.ascontiguousarray(np.transpose(sc.class_maps, (0, 2, 1)), np.float32).tofile(os.path.join(d, "maps.bin"))
    np.ascontiguousarray(sc.class_mask.T, np.uint8).tofile(os.path.join(d, "mask.bin"))
    pcl = np.zeros((len(sc.pts), 8), np.float32)
    pcl[:, :3] = sc.pts[:, :3]
    pcl[:, 4] = sc.pts[:, 3]
    pcl.tofile(os.path.join(d, "pts.bin"))
    sc.states.tofile(os.path.join(d, "states.bin"))
    env = dict(os.environ, TDR_FACADE_BENCH=str(steps))
    r = subprocess.run([exe, d], capture_output=True, text=True, env=env)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr)
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
