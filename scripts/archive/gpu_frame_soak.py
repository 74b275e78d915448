"""Soak of the one-call callback (tdlo_tracker_frame_from_depth): many frames of a drifting rope at 640 x 480 with a hidden stretch now and then, against the
three separate calls on a second context (bit-equal, looked at every 500 frames), device memory and route counters at the end.   usage: python scripts/gpu_frame_soak.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth


def main(n_frames):
    import torch
    P = synth.LAUNCH_PARAMS
    M = 30
    a, b = B.Context(device=0, timing=False), B.Context(device=0, timing=False)
    scenes = [synth.depth_scene(M, config=9, frame=f) for f in range(8)]
    depth0, mask0, cam, Y0 = scenes[0]
    coord = synth.geodesic_coord(Y0)
    args = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    targs = (M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 30, P["tol"], P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
    ta, tb = B.trackdlo(*targs, ctx=a), B.trackdlo(*targs, ctx=b)
    for t in (ta, tb):
        t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord)
    da, ma = a.image_buffers(*depth0.shape)
    free0 = None
    t0 = time.perf_counter(); checked = 0
    for fr in range(n_frames):
        if fr == 100: free0 = torch.cuda.mem_get_info(0)[0]          # (after the first frames: every workspace exists)
        depth, mask, _, _ = scenes[fr % 8]
        hide = fr % 97 == 13
        da[:] = depth; ma[:] = mask
        if hide: ma[:, 300:330] = 0
        ta.frame_from_depth(da, ma, *args, 0.008, 0.06)
        # the three separate calls on the second context, frame by frame: the same bits (looked at every 500 frames)
        b.depth_to_cloud(0, da, ma, *args, 0.008, fetch=False)
        _, vb, eb = b.visibility_prepass(0, tb.get_tracking_result(), P["visibility_threshold"], 0.06, coord)
        tb.tracking_step(None, vb, eb)
        if fr % 500 == 499:
            assert np.array_equal(ta.get_tracking_result(), tb.get_tracking_result()) and ta.get_sigma2() == tb.get_sigma2(), fr
            checked += 1
    dt = time.perf_counter() - t0
    free1 = torch.cuda.mem_get_info(0)[0]
    print(f"{n_frames} frames in {dt:.1f} s (both forms, image copies included), {checked} bit-equality checks against the three separate calls; device memory {free0 - free1:+d} bytes; "
          f"pre-pass rides {a.cloud_vis_rides()}, cloud routes {a.cloud_route_counts()}, tracker routes {a.route_counts()}; nodes finite: {bool(np.all(np.isfinite(ta.get_tracking_result())))}")
    a.close(); b.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30000)
