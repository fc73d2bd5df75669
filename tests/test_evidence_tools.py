"""The tools that turn rocprofv3 output into the bench line's roofline evidence (tools/isa_cost.py, bench.source_hash): their
parsing, on CPU.  A silent fallback here changes a committed number (round 3: function labels with a trailing comment were not
recognised and every kernel was priced at the 2.9-cycle default)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

ISA = """\t.text
\t.protected\t_ZN5crtfx6k_demoEv ; -- Begin function _ZN5crtfx6k_demoEv
\t.globl\t_ZN5crtfx6k_demoEv
\t.type\t_ZN5crtfx6k_demoEv,@function
_ZN5crtfx6k_demoEv:                     ; @_ZN5crtfx6k_demoEv
; %bb.0:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tv_fma_f32 v0, v1, v2, v3
\tv_pk_fma_f32 v[0:1], v[2:3], s[0:1], v[0:1]
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\tv_cvt_f64_f32_e32 v[4:5], v0
\tds_read_b32 v6, v7
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
\t.size\t_ZN5crtfx6k_demoEv, .Lfunc_end0-_ZN5crtfx6k_demoEv
_ZN5crtfx7k_emptyEv:
\ts_endpgm
.Lfunc_end1:
"""


def test_static_mix_reads_function_labels_with_trailing_comments(tmp_path):
    import isa_cost
    p = tmp_path / "demo.s"
    p.write_text(ISA)
    mix = isa_cost.static_mix(str(p))
    assert set(mix) == {"_ZN5crtfx6k_demoEv", "_ZN5crtfx7k_emptyEv"}
    n, cyc = mix["_ZN5crtfx6k_demoEv"]
    assert n == 3                                   # the three v_ instructions; local labels do not start a function
    assert abs(cyc - (isa_cost.cost("v_fma_f32") + isa_cost.cost("v_pk_fma_f32") + isa_cost.cost("v_cvt_f64_f32_e32"))) < 1e-9
    assert mix["_ZN5crtfx7k_emptyEv"] == (0, 0.0)


def test_every_priced_mnemonic_has_a_positive_cost():
    import isa_cost
    for m in ("v_fma_f32", "v_pk_fma_f32", "v_add_f64", "v_cvt_pk_u8_f32", "v_readlane_b32", "v_mov_b32_e32", "v_unknown_op"):
        assert isa_cost.cost(m) > 0
    assert isa_cost.cost("s_waitcnt") == 0.0 and isa_cost.cost("ds_read_b32") == 0.0


def test_source_hash_covers_every_kernel_source():
    """bench.py nulls profiles/traffic.json / valu.json figures measured on other sources: the hash must move with any file
    the library is built from."""
    import bench
    from pythoncrt_amd import _lib
    names = {os.path.basename(p) for p in _lib.SOURCES}
    for f in os.listdir(_lib.CSRC):
        if f.endswith((".hip", ".h")):
            assert f in names, f"{f} is compiled into libcrtfx.so but not part of SOURCES / the evidence hash"
    assert "crtfx.h" in names
    h = bench.source_hash()
    assert len(h) == 16 and int(h, 16) >= 0


def test_bench_ceiling_and_telemetry_degrade_gracefully():
    """bench.py's round-4 additions on a box without a GPU: the committed Infinity-Cache-resident ceiling of another box is a reference figure
    (4K groups, 1080p groups; nothing for other sizes), the one in `roofline` is measured in the run or absent, and the per-rank GPU telemetry turns into nulls with a note instead of raising
    when rocm_smi / the device is unavailable."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    a = bench.mall_ceiling_committed(2.0, 2160)
    assert a and a["frames"] == 2 and 6000 < a["gbs"] < 8000 and a["source"].startswith("profiles/r04_mall_copy.txt")
    b = bench.mall_ceiling_committed(5.0, 1080)
    assert b and b["frames"] == 5 and b["us_per_group"] > 0 and "1080p" in b["source"]
    assert bench.mall_ceiling_committed(1.0, 4320) is None and bench.mall_ceiling_committed(7.0, 2160) is None
    # the figure under `roofline` is measured in the run (tools/ubench/mall_copy.hip as a child process): without the program (or without a
    # GPU: it then exits non-zero) there is NO figure, never the committed one of another box
    rows = bench._mall_rows("2 frame(s) (199 MB float32 just written) x3          36.5 us    6814 GB/s (source + uint8 out)\n"
                            "2 frame(s) (199 MB float32 just written) tap4        40.8 us    6095 GB/s (source + uint8 out)\n", 2)
    assert rows["x3"] == {"us_per_group": 36.5, "gbs": 6814.0} and rows["tap4"]["us_per_group"] == 40.8
    assert bench.mall_ceiling(2.0, 2160, 3840) is None
    import torch
    t = bench.GpuTelemetry(torch.device("cpu"))
    t.start()
    out = t.stop()
    assert out["sclk_mhz_mean"] is None and out["power_w_mean"] is None and out["samples"] == 0


def test_bench_preflight_memory_estimate():
    """bench.preflight_need_bytes: the default workloads fit one 288 GB MI355X with room to spare, an oversized batch does not, and the
    estimate covers at least the buffers bench.py visibly allocates (resident frames + output slots)."""
    import bench
    gib = 2 ** 30
    cases = {"4K": (2160, 3840, 1920, 1, 0.0, 1, 0), "1080p persistence sharded": (1080, 1920, 4096, 1, 0.5, 2, 26), "8K half": (4320, 7680, 384, 2, 0.0, 1, 0)}
    for name, c in cases.items():
        need = bench.preflight_need_bytes(*c)
        h, w, b, eb, p, slots, keep = c
        assert need >= b * h * w * 3 * eb * (1 + slots), name
        assert need < 230 * gib, (name, need / gib)
    assert bench.preflight_need_bytes(2160, 3840, 400000, 1, 0.0, 1, 0) > 288 * gib
    assert bench.preflight_need_bytes(1080, 1920, 64, 1, 0.5, 2, 26) > bench.preflight_need_bytes(1080, 1920, 64, 1, 0.0, 2, 0)


def test_committed_bench_lines_of_the_newest_round_carry_their_counters():
    """Every committed per-config bench line of the newest round (profiles/rNN_z*_bench.json) was taken on the sources in the tree AFTER the
    PMC passes of those sources had been summarised: `roofline.traffic` is a number, its provenance names the build's hash (not `stale`), the
    bound is counter-derived and the plan is recorded.  Round 4 committed four lines that said `stale: true` / `bound: "hbm"`: the staleness
    guard had worked, nobody had re-taken the lines.  (A change to any hashed source — pythoncrt_amd/csrc/*, include/crtfx.h — means: run
    tools/collect_profiles.sh + summarise_profiles.py per config, then `bench.py --config N` again, and commit the five lines.)"""
    import glob
    import json
    import re
    import bench
    rounds = sorted({int(m.group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_z*_bench.json"))
                     for m in [re.match(r"r(\d+)_z", os.path.basename(f))] if m})
    assert rounds, "no committed bench lines"
    newest = rounds[-1]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r{newest:02d}_z*_bench.json")))
    assert len(files) >= 5, files                      # the headline + configs 0, 2, 4, 5
    now = bench.source_hash()
    hashes = {(json.load(open(f))["roofline"].get("traffic_source") or {}).get("source_hash") for f in files}
    if hashes != {now} and os.environ.get("CRTFX_EVIDENCE_GATE") != "1":
        # a kernel / header edit since the last evidence round: a release gate, not a unit test (round-5 advisor finding: the CPU suite must not
        # stay red from the first csrc edit of a round until its last GPU call).  CRTFX_EVIDENCE_GATE=1 (tools/evidence_round.sh sets it for its
        # closing check) makes this a failure again.
        import pytest
        pytest.xfail(f"committed bench lines were measured on sources {sorted(h or 'none' for h in hashes)}, the tree is {now}: "
                     f"re-run tools/evidence_round.sh (GPU box) and commit profiles/")
    for f in files:
        d = json.load(open(f))
        r = d["roofline"]
        src = r.get("traffic_source") or {}
        assert not src.get("stale"), (os.path.basename(f), src)
        assert src.get("source_hash") == now, (os.path.basename(f), src.get("source_hash"), now)
        assert isinstance(r.get("traffic"), int) and r["traffic"] > 0 and r.get("fabric_frac"), os.path.basename(f)
        assert r.get("valu") and r.get("bound_evidence", {}).get("fractions_of_each_limit"), os.path.basename(f)      # a counter-derived bound, not the "hbm" default
        assert r.get("plan") and d.get("config", {}).get("workload"), os.path.basename(f)
        if r.get("mall_ceiling") is not None:
            assert "this box" in r["mall_ceiling"]["source"], os.path.basename(f)       # measured in the run, never a committed figure


# ---- the resources the four-blocks-per-CU design hangs on (round 6) -------------------------------------------------------------------------
# k_phosphor_ct is planned for FOUR resident 256-thread blocks per CU (plan_grid simulates 4 x 256 block slots): that needs <= 128 VGPRs per
# lane (512 / 4), no spills (scratch traffic in the trip loop) and <= 40 960 bytes of LDS per block (160 KB / 4).  A compiler bump, a flag or one
# more live value that pushes a build over either limit costs 12 - 22 % (profiles/r03_ct_ablation.txt, E: three blocks per CU) with every
# parity test and the plan string unchanged.  The figures are read from the code objects inside the built libcrtfx.so (tools/kernel_resources.py)
# and from crtfx_kernel_lds_bytes — no GPU.  Shown to guard (profiles/r06_resource_guard.txt): a library built with -DCT_WAVES=5 or
# -DCT_RING_ROWS_N=64 fails `check_resources`.

CT_PINNED = {                       # (radius, pix): (VGPRs, LDS bytes) of the builds the BASELINE configs land on (tests/test_plan_gpu.py)
    (9, 0): (110, 39712),           # configs[2] 4K: k_phosphor_ct<9,u8>
    (9, 1): (118, 35616),           # configs[4] 8K half: k_phosphor_ct<9,half>
    (4, 0): (94, 38176),            # configs[1], [3] 1080p: k_phosphor_ct<4,u8>
}
# the other kernels of the configs' plans: VGPR ceiling = the occupancy step the committed build sits under (256-thread blocks: waves per SIMD =
# 512 // VGPRs rounded up to 8), static LDS bytes
OTHER_PINNED = {
    "crtfx::k_warp_lean<true, 0, 0, 4, false, 2, false, true>": (72, 0),      # configs 2, 3: f64, no blend, u8, 4 rows, plain (67 VGPRs: 7 waves)
    "crtfx::k_warp_lean<true, 1, 0, 2, false, 1, true, true>": (80, 0),       # config 4: persistence run, 2 rows, plain (76: 6 waves)
    "crtfx::k_warp_lean<true, 0, 1, 4, false, 2, false, true>": (80, 0),      # config 5: half rows (74: 6 waves)
    "crtfx::k_point_fused_seq<16821680u, 0, 1>": (96, 8224),                  # the reference CLI's defaults: fast bloom + pixelate, u8, render blend (90; two 512-thread blocks per CU
                                                                              # need <= 128; + 43.5 KB of dynamic LDS per block for the eight frames' half-resolution tiles)
    "crtfx::k_point_lean_seq<16821680u, 0, 1>": (128, 8224),                  # ... its two-launch form (odd frame sizes, NO_FUSED_HALF): 106
    "crtfx::k_half_group<16821680u, 0>": (32, 0),                             # ... and that form's half-resolution bloom source (24)
}


def check_resources(lib_path):
    """Problems (strings) with the register / spill / LDS figures of the library at lib_path; [] = every pinned figure holds."""
    import ctypes
    import kernel_resources
    res = kernel_resources.resources(lib_path)
    lib = ctypes.CDLL(lib_path)
    lib.crtfx_kernel_lds_bytes.restype = ctypes.c_int
    lib.crtfx_kernel_lds_bytes.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    bad = []
    for pix in (0, 1):
        for radius in range(1, 16):
            k = res.get(f"crtfx::k_phosphor_ct<{radius}, {pix}>")
            if k is None:
                bad.append(f"k_phosphor_ct<{radius},{pix}> is not in the library")
                continue
            lds = lib.crtfx_kernel_lds_bytes(b"k_phosphor_ct", radius, pix)
            if not (0 < lds <= 40960):
                bad.append(f"k_phosphor_ct<{radius},{pix}>: {lds} bytes of LDS > 40960 (three blocks per CU)")
            if k["vgpr_count"] + k["agpr_count"] > 128:
                bad.append(f"k_phosphor_ct<{radius},{pix}>: {k['vgpr_count']} + {k['agpr_count']} VGPRs > 128 (fewer than four waves per SIMD)")
            if k["group_segment_fixed_size"] != 0:
                bad.append(f"k_phosphor_ct<{radius},{pix}>: static LDS {k['group_segment_fixed_size']} on top of the dynamic block")
            if radius <= 12 and (k["vgpr_spill_count"] or k["sgpr_spill_count"] or k["private_segment_fixed_size"]):
                bad.append(f"k_phosphor_ct<{radius},{pix}>: spills ({k['vgpr_spill_count']} VGPR, {k['sgpr_spill_count']} SGPR, "
                           f"{k['private_segment_fixed_size']} bytes of scratch) at a radius that had none")
            if radius > 12 and k["vgpr_spill_count"] > 24:
                bad.append(f"k_phosphor_ct<{radius},{pix}>: {k['vgpr_spill_count']} VGPRs spilled (committed: <= 23, profiles/r05_half_sigma.txt)")
    for (radius, pix), (vg, lds_b) in CT_PINNED.items():
        k = res.get(f"crtfx::k_phosphor_ct<{radius}, {pix}>")
        if k is None:
            continue
        if k["vgpr_count"] > vg + 6:            # a few registers of compiler noise are fine, a jump towards the 128 limit is a change to look at
            bad.append(f"k_phosphor_ct<{radius},{pix}>: {k['vgpr_count']} VGPRs, committed {vg}")
        if lib.crtfx_kernel_lds_bytes(b"k_phosphor_ct", radius, pix) != lds_b:
            bad.append(f"k_phosphor_ct<{radius},{pix}>: LDS {lib.crtfx_kernel_lds_bytes(b'k_phosphor_ct', radius, pix)} bytes, committed {lds_b}")
    for name, (vmax, lds_b) in OTHER_PINNED.items():
        k = res.get(name)
        if k is None:
            bad.append(f"{name} is not in the library")
            continue
        if k["vgpr_count"] + k["agpr_count"] > vmax:
            bad.append(f"{name}: {k['vgpr_count']} VGPRs > {vmax} (one occupancy step down)")
        if k["vgpr_spill_count"] or k["sgpr_spill_count"] or k["private_segment_fixed_size"]:
            bad.append(f"{name}: spills / scratch ({k['vgpr_spill_count']}, {k['sgpr_spill_count']}, {k['private_segment_fixed_size']} B)")
        if k["group_segment_fixed_size"] != lds_b:
            bad.append(f"{name}: static LDS {k['group_segment_fixed_size']} bytes, committed {lds_b}")
    return bad


def test_headline_kernels_keep_their_registers_and_lds():
    from pythoncrt_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    bad = check_resources(_lib.LIB_PATH)
    assert not bad, "\n".join(bad)


def test_kernel_lds_query_arguments():
    from pythoncrt_amd import _lib
    lib = _lib.load()
    assert lib.crtfx_kernel_lds_bytes(b"k_phosphor_ct", 16, 0) == -3 and lib.crtfx_kernel_lds_bytes(b"k_phosphor_ct", 0, 0) == -3
    assert lib.crtfx_kernel_lds_bytes(b"k_phosphor_ct", 9, 7) == -1 and lib.crtfx_kernel_lds_bytes(None, 9, 0) == -1
    assert lib.crtfx_kernel_lds_bytes(b"k_nothing", 9, 0) == -3 and lib.crtfx_kernel_lds_bytes(b"k_phosphor_cc", 20, 0) > 40960


def test_pmc_vmem_aggregator_refuses_a_missing_pass(tmp_path):
    """tools/pmc_vmem_aggregate.py: a pass without a CSV, or a kernel without one of a pass's counters, is an error and NO JSON is written
    (round 5: the `tc` pass died inside rocprofv3 and the script wrote a JSON without its counters)."""
    import pmc_vmem_aggregate as agg
    passes = ["ta|TA_BUSY GRBM", "tc1|TA_STALL"]

    def write(name, rows):
        d = tmp_path / f"t_{name}" / "host" / "1"
        d.mkdir(parents=True, exist_ok=True)
        with open(d / "1_counter_collection.csv", "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for r in rows:
                f.write(",".join(str(x) for x in r) + "\n")

    write("ta", [('"void crtfx::k_a<1>(int)"', "TA_BUSY", 10), ('"void crtfx::k_a<1>(int)"', "TA_BUSY", 30), ('"void crtfx::k_a<1>(int)"', "GRBM", 5),
                 ("other_kernel", "TA_BUSY", 99)])
    assert agg.main(["x", str(tmp_path), "t"] + passes) == 2 and not (tmp_path / "t_vmem.json").exists()          # pass tc1 missing
    write("tc1", [('"void crtfx::k_b<2>(int)"', "TA_STALL", 7)])
    assert agg.main(["x", str(tmp_path), "t"] + passes) == 2 and not (tmp_path / "t_vmem.json").exists()          # k_a lacks TA_STALL, k_b the ta pass
    write("tc1", [('"void crtfx::k_a<1>(int)"', "TA_STALL", 7)])
    assert agg.main(["x", str(tmp_path), "t"] + passes) == 0
    import json
    out = json.load(open(tmp_path / "t_vmem.json"))
    assert out == {"crtfx::k_a<1>": {"TA_BUSY": 20.0, "GRBM": 5.0, "TA_STALL": 7.0}}


def test_bench_keeps_stdout_to_one_json_line_around_the_rccl_banner():
    """bench.stdout_to_stderr: whatever a library prints on file descriptor 1 inside the block (RCCL's five-line start-up banner, measured in round 6)
    lands on stderr; Python's own stdout works again afterwards.  Also: --force-dist is a world-size-1 rehearsal only."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "print('before', flush=True)\n"
            "with bench.stdout_to_stderr():\n"
            "    os.write(1, b'RCCL version : banner\\n')\n"
            "    print('python inside', flush=True)\n"
            "print('after', flush=True)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout.split() == ["before", "after"], r.stdout
    assert "RCCL version : banner" in r.stderr and "python inside" in r.stderr
    import bench
    a = bench.parse_args(["--gpus", "1", "--force-dist", "--config", "4"])
    assert a.force_dist and a.gpus == 1 and a.config == 4


def _synthetic_line(n, value, clk=2250.0, pw=1320.0, own=None, chain=0.0194, sched=None, workload="BASELINE configs[3]: 1920x1080 chain, persistence 0.5, u8 in/out"):
    own = own or [value / n] * n
    return {"metric": "1080p frames/sec (whole node)", "value": value, "n_gpus": n, "config": {"workload": workload},
            "dist": {"world_size_seen": n, "hop_schedule": "synchronous" if sched else None,
                     "per_rank": [{"rank": r, "frames_per_s_own_clock": own[r], "chain_ms_per_frame": chain,
                                   "gpu": {"sclk_mhz_mean": clk if not isinstance(clk, list) else clk[r], "power_w_mean": pw}} for r in range(n)]},
            **({"shard_schedule": sched} if sched else {})}


def test_scale_report_reads_the_cause_from_the_lines(tmp_path, capsys):
    """tools/scale_report.py — the 8-GPU run sheet as a program: from bench.py lines at N = 1 and N = 8 it states the speed-up
    against north_star's >= 7x and, under target, the cause the per-rank fields point at: a node power cap (clocks below the 1-GPU line's), the
    persistence hop (hop + fix-up share of a round), a straggling host (one rank slow on its own clock with equal kernel time)."""
    import json
    import scale_report as sr
    base = _synthetic_line(1, 51600.0)
    sched_ok = {"scan_us": 79000.0, "hop_stall_us": 160.0, "fixup_us": 150.0, "hop_plus_fixup_share": 0.0039}
    cases = {
        "healthy": (_synthetic_line(8, 8 * 51600.0 * 0.97, sched=sched_ok), 0, ["7.76x", "MET"], ["->"]),
        "capped": (_synthetic_line(8, 8 * 51600.0 * 0.80, clk=1850.0, pw=1050.0, sched=sched_ok), 1, ["NOT MET", "power / thermal cap"], []),
        "hop": (_synthetic_line(8, 8 * 51600.0 * 0.84, sched={"scan_us": 79000.0, "hop_stall_us": 14000.0, "fixup_us": 150.0, "hop_plus_fixup_share": 0.152}), 1,
                ["NOT MET", "CRTFX_SHARD_OVERLAP=1"], ["power / thermal cap"]),
        "straggler": (_synthetic_line(8, 8 * 43000.0, own=[51000.0] * 7 + [43000.0], sched=sched_ok), 1, ["NOT MET", "straggler"], ["power / thermal cap"]),
    }
    for name, (line, rc, must, must_not) in cases.items():
        p1, p8 = tmp_path / f"{name}_1.json", tmp_path / f"{name}_8.json"
        p1.write_text("RCCL version : banner line\n" + json.dumps(base) + "\n")          # other lines around the JSON line are ignored
        p8.write_text(json.dumps({"rc": 0, "parsed": line}))                              # ... and a driver record with the line under "parsed" is accepted
        assert sr.main(["scale_report.py", str(p8), str(p1)]) == rc, name
        out = capsys.readouterr().out
        for m in must:
            assert m in out, (name, m, out)
        for m in must_not:
            assert m not in out, (name, m, out)
    # the committed rehearsal (five gloo ranks on ONE GPU) parses: every field the run sheet names is in the line
    reh = json.load(open(os.path.join(ROOT, "profiles", "r06_five_rank_rehearsal.json")))
    verdict, text = sr.diagnose(json.load(open(os.path.join(ROOT, "profiles", "r06_z_c4_bench.json"))), reh["config4"])
    assert verdict == "UNDER TARGET" and any("persistence hop (overlapped)" in t for t in text) and any("shader clock per rank" in t for t in text)
