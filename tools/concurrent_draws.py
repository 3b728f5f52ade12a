#!/usr/bin/env python3
"""tools/concurrent_draws.py -- what kernel instances in flight buy the blur sub-frames of a clip frame (development aid, GPU box).

    python tools/concurrent_draws.py [--precompile]

For each (scene, size): ROUNDS x 4 sub-frames whose uniforms move from one to the next (what `portal-amd render --motion-blur-frames 4` issues),
with "concurrent_draws" 1 (one kernel instance: every launch waits for the previous one's uniform block), 2 and 4, and as ONE launch
with grid.z = 4 (FLAG_SLICES: one uniform block per slice in a device buffer); wall time from the first
launch to a device synchronize, per sub-frame, and a hash over all frames that must not depend on K."""
import hashlib, json, os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

CASES = [("monoportal", 1920, 1080, 20, 1), ("monoportal", 1280, 720, 20, 1), ("portal_in_portal", 1920, 1080, 40, 1), ("portal_in_portal", 3840, 2160, 40, 4)]
ROUNDS, N = 24, 4

if __name__ == "__main__":
    pre = "--precompile" in sys.argv
    for name, w, h, depth, aa in CASES:
        for lanes in (1, 2, 4, "one launch"):
            scene = pa.Scene.from_file(pa.scene_path(name))
            sliced = lanes == "one launch"
            r = pa.SceneRenderer(scene, device=-1 if pre else 0, flags=pa.FLAG_SPECIALIZE_STATIC | (pa.FLAG_SLICES if sliced else 0))
            if pre:
                if sliced:
                    break
                continue
            if sliced:
                r.set_option("render_depth", depth)
                r.set_option("aa_count", aa)
                big = pa.device_alloc(N * w * h * 4, 0)
                frame = pa.Frame(w, h, 0, 1)
                digest = hashlib.sha1()

                def batch(k0):
                    for j in range(N):
                        k = k0 + j
                        r.set_camera((0.01 * (k % 7), 0.1, -0.3), 0.9 + 0.01 * (k % 11), 1.2, 3.1)
                        r.set_option("aa_start", j)
                        r.stage_slice(frame, j)
                    r.draw_slices(frame, N, out_rgba8=big, slice_pixels=w * h)

                batch(0)
                digest.update(pa.device_download(big, N * w * h * 4).tobytes())
                t0 = time.perf_counter()
                for rnd in range(ROUNDS):
                    batch(rnd * N)
                pa.device_download(big, 16)
                dt = time.perf_counter() - t0
                print(json.dumps({"scene": name, "size": f"{w}x{h}", "aa": aa, "concurrent_draws": "one launch, grid.z = 4", "ms_per_subframe": round(dt / (ROUNDS * N) * 1e3, 4),
                                  "sha": digest.hexdigest()[:10]}), flush=True)
                pa.device_free(big)
                continue
            r.set_option("render_depth", depth)
            r.set_option("aa_count", aa)
            r.set_option("concurrent_draws", lanes)
            bufs = [pa.device_alloc(w * h * 4, 0) for _ in range(N)]
            frame = pa.Frame(w, h, 0, 1)
            digest = hashlib.sha1()

            def one_round(k0, check):
                for j in range(N):
                    k = k0 + j
                    r.set_camera((0.01 * (k % 7), 0.1, -0.3), 0.9 + 0.01 * (k % 11), 1.2, 3.1)
                    r.set_option("aa_start", j)
                    r.draw_device(frame, out_rgba8=bufs[j])
                r.join()
                if check:
                    for b in bufs:
                        digest.update(pa.device_download(b, w * h * 4).tobytes())

            one_round(0, True)  # warm-up (clones are made here) + the frames that are hashed
            pa.device_download(bufs[0], 16)
            t0 = time.perf_counter()
            for rnd in range(ROUNDS):
                one_round(rnd * N, False)
            pa.device_download(bufs[N - 1], 16)  # a blocking copy on the default stream: everything before it has finished
            dt = time.perf_counter() - t0
            print(json.dumps({"scene": name, "size": f"{w}x{h}", "aa": aa, "concurrent_draws": lanes, "ms_per_subframe": round(dt / (ROUNDS * N) * 1e3, 4),
                              "sha": digest.hexdigest()[:10]}), flush=True)
            for b in bufs:
                pa.device_free(b)
