"""CPU: bench.py's `roofline.traffic` constant is the committed PMC measurement it cites (the number
must not go stale silently when the kernel or its profile changes)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_constant_matches_committed_pmc_summary():
    import bench
    nbytes, src = bench.TRAFFIC["rgba8"]
    path = os.path.join(ROOT, src.split(":")[0])
    assert os.path.exists(path), path
    vals = {}
    for ln in open(path):
        m = re.match(r"\s*(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE) n=\d+ avg=([0-9.]+)", ln)
        if m and m.group(1).startswith("dwt53_fwd_rgba8_wg_kernel"):
            vals[m.group(2)] = float(m.group(3))
    assert set(vals) == {"FETCH_SIZE", "WRITE_SIZE"}
    # gfx950: FETCH_SIZE reads half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM); both in KiB
    measured = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    assert abs(measured - nbytes) <= 0.001 * nbytes, (measured, nbytes)
    alg = 3840 * 2160 * 16                                  # one RGBA8 dword in, three int32 out per pixel
    assert 1.0 <= nbytes / alg < 1.03                       # halo re-reads are L2 hits: traffic ~= algorithmic bytes


def test_c3_traffic_constant_matches_committed_pmc_summary():
    import bench_extra
    nbytes, src = bench_extra.TRAFFIC["c3"]
    path = os.path.join(ROOT, src.split(":")[0])
    assert os.path.exists(path), path
    vals = {}
    for ln in open(path):
        m = re.match(r"\s*(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE) n=\d+ avg=([0-9.]+)", ln)
        if m and m.group(1).startswith("dwt97_fwd_rgb_wg_kernel<8, 1, 7, 0>"):        # level 0 (<..., 1>: the deeper levels)
            vals[m.group(2)] = float(m.group(3))
    assert set(vals) == {"FETCH_SIZE", "WRITE_SIZE"}
    measured = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    assert abs(measured - nbytes) <= 0.001 * nbytes, (measured, nbytes)
    # the PMC pass ran the configuration's default: a batch of four frames per launch (the constant and the launch it
    # is quoted for must be the same launch)
    assert bench_extra.TRAFFIC_BATCH["c3"] == bench_extra.CONFIGS["c3"]["batch"] == 4
    assert "--batch" not in open(path).readline()           # (the summary's command line: no override of the batch)
    alg = 3840 * 2160 * 27 * 4                              # 12 B in; 3/4 int32 + 1/4 float64 out per sample, three components, four frames
    assert 1.0 <= nbytes / alg < 1.03


def test_pipeline_traffic_constant_is_the_sum_over_the_steps_kernels():
    """bench.PIPELINE_TRAFFIC (the `pipeline` object of the JSON line): (2 x FETCH_SIZE + WRITE_SIZE) added over the nine
    kernels of a C2 step, from the committed summary it cites"""
    import bench
    nbytes, src = bench.PIPELINE_TRAFFIC
    path = os.path.join(ROOT, src.split(":")[0])
    assert os.path.exists(path), path
    vals = {}
    for ln in open(path):
        m = re.match(r"\s*(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE) n=\d+ avg=([0-9.]+)", ln)
        if not m:
            continue
        for k in bench.PIPELINE_KERNELS:
            if k in m.group(1):
                vals[(k, m.group(2))] = float(m.group(3))
    assert len(vals) == 2 * len(bench.PIPELINE_KERNELS), sorted(vals)
    measured = sum((2 if c == "FETCH_SIZE" else 1) * v for (_, c), v in vals.items()) * 1024
    assert abs(measured - nbytes) <= 0.001 * nbytes, (measured, nbytes)
