"""-m gpu: the hot path at BASELINE.json's full size (configs[1]: FLUX.1-Kontext dims, 1024^2 -> L = L_c = 4096, T = 512,
11.9 B synthetic parameters) through size-independent properties - the oracle cannot run this size in seconds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flux_1024_full_size_properties(golden):
    import bench as B
    from regione_amd import RegionEHelper, synth, ops
    from regione_amd.harness import flux as HF
    from regione_amd.FluxKontext.inplace import gamma
    from tools.run_configs import weights_stream, make_box, expected_ids
    dev = torch.device("cuda", 0)
    cfg = synth.FluxConfig()
    pipe = HF.FluxKontextPipeline(HF.FluxTransformer2DModel(cfg, dev).load_state_dict_stream(weights_stream(cfg, dev, 42)))
    h = w = 64
    L, T = h * w, 512
    lat, img, prompt, pooled = [t.to(dev) for t in synth.make_edit_inputs(h, w, T, cfg, seed=110)]
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.88, cache_threshold=0.04)
    helper.enable()
    box = make_box(h, w, 0.25)
    B.install_region_injection(pipe, h, w, box, img[0:1], seed=7)
    trace = {}
    stepped = []                                                   # scheduler outputs BEFORE the manager compacts / restores them
    out = pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=1024, width=1024, latents=lat,
               guidance_scale=2.5, return_dict=False, trace=trace,
               callback_on_step_end=lambda p, i, t, kw: stepped.append(kw["latents"].clone()))[0]
    M = pipe._regione_manager
    kinds = "".join(trace["kind"])
    # (1) the F/R/C plan is the one the REFERENCE's loop executes at this sequence length (tests/golden/loop_plan_64.npz)
    assert kinds == "".join(golden("loop_plan_64")["kinds"].tolist())
    # (2) partition: edited ids = the constructed region, ascending; edited + unedited = a partition of all tokens
    e, u = M.edited_ids.squeeze(0).cpu(), M.unedited_ids.squeeze(0).cpu()
    assert torch.equal(e, expected_ids(h, w, box)) and bool((e[1:] > e[:-1]).all()) and bool((u[1:] > u[:-1]).all())
    assert torch.equal(torch.sort(torch.cat([e, u])).values, torch.arange(L))
    # (3) sequence lengths follow the stage machine: full until the partition, K_e inside RAGS, full again for the tail
    #     (same full / compacted pattern as the reference run at this length, with this run's K_e)
    lens = [x.shape[1] for x in trace["latents"]]
    K = e.numel()
    ref_len = golden("loop_plan_64")["len"].tolist()
    assert lens == [L if n == L else K for n in ref_len] and 0 < K < L
    # (4) a cache-served step is EXACTLY the cached velocity times its decay ratio (bit for bit, at any size)
    ts = pipe.scheduler.timesteps.float().cpu()
    last = None
    checked = 0
    for i, k in enumerate(kinds):
        v = trace["noise_pred"][i]
        if k == "C":
            ratio = gamma[i - 1] * (1 + (ts[i] - ts[i - 1]) / 1000)
            src = last if last.shape[1] == v.shape[1] else ops.gather_rows(last, M.edited_ids)
            assert torch.equal(v, ops.avd_apply(src, float(ratio))), i
            checked += 1
        last = v if k != "C" else (last if last.shape[1] == v.shape[1] else ops.gather_rows(last, M.edited_ids))
    assert checked == kinds.count("C") == 14
    # (5) everything finite, and the region steps only ever touched the edited rows: between the partition and the
    #     refresh the unedited rows of the reassembled latent do not move
    assert torch.isfinite(out.float()).all() and out.shape == (1, L, 64)
    ei, ui = M.edited_ids.squeeze(0), M.unedited_ids.squeeze(0)
    held, compactions, restores = None, 0, 0
    for before, after in zip(stepped, trace["latents"]):
        if before.shape[1] == L and after.shape[1] == K:                        # partition / re-compaction after a refresh
            held = before[0, ui]
            assert torch.equal(after[0], before[0, ei])
            compactions += 1
        elif before.shape[1] == K and after.shape[1] == L:                      # restore before a refresh / the tail
            assert torch.equal(after[0, ui], held), "unedited rows moved while the loop ran on the edited rows only"
            assert torch.equal(after[0, ei], before[0])
            restores += 1
        else:
            assert torch.equal(after, before)
    assert compactions == restores == 2 and len(stepped) == 28                  # partition + one forced refresh (step 16)


def test_qwen_1024_full_size_plan_and_partition(golden):
    """BASELINE configs[2] at full size: Qwen-Image-Edit dims (60 double-stream blocks, text width 3584, 20 B synthetic
    parameters), 1024^2, true CFG with text lengths 512 / 384, two K/V caches per layer: reference plan at L = 4096,
    constructed region recovered exactly, finite output."""
    import bench as B
    from regione_amd import RegionEHelper, synth
    from regione_amd.harness import qwen as HQ
    from tools.run_configs import weights_stream, make_box, expected_ids
    dev = torch.device("cuda", 0)
    cfg = synth.FluxConfig(**synth.QWEN)
    pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, dev).load_state_dict_stream(weights_stream(cfg, dev, 42)))
    h = w = 64
    L = h * w
    lat, img, prompt, _ = [t.to(dev) if t is not None else None for t in synth.make_edit_inputs(h, w, 512, cfg, seed=110)]
    _, _, nprompt, _ = synth.make_edit_inputs(h, w, 384, cfg, seed=111)
    helper = RegionEHelper(pipe)
    helper.set_params()
    helper.enable()
    box = make_box(h, w, 0.25)
    B.install_region_injection(pipe, h, w, box, img[0:1], seed=7)
    trace = {}
    out = pipe(image=img, prompt_embeds=prompt, negative_prompt_embeds=nprompt.to(dev), height=1024, width=1024, latents=lat,
               true_cfg_scale=4.0, return_dict=False, trace=trace)[0]
    assert "".join(trace["kind"]) == "".join(golden("qwen_plan_64")["kinds"].tolist())
    assert torch.equal(pipe._regione_manager.edited_ids.squeeze(0).cpu(), expected_ids(h, w, box))
    assert torch.isfinite(out.float()).all() and out.shape == (1, L, 64)
    del pipe
    torch.cuda.empty_cache()


@pytest.mark.parametrize("extra", [[], ["--true-cfg", "6.0"]])
def test_bench_two_ranks_share_the_gpu(extra):
    """bench.py's N > 1 path on a one-GPU box: two ranks (gloo, both on cuda:0, toy trunk) launched exactly like the driver
    launches N ranks.  One JSON line from rank 0; per-rank K_e cycles 5 % / 15 %; `--true-cfg 6.0` = BASELINE configs[3]."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + (os.getpid() % 400) + (7 if extra else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--toy", "--share-gpu",
           "--dist-backend", "gloo", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                               # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "steps/s"
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and [p["edit_frac"] for p in d["per_rank"]] == [0.05, 0.15]
    assert all(p["edit_s"] > 0 for p in d["per_rank"]) and d["per_rank"][0]["K_e"] < d["per_rank"][1]["K_e"]
    assert ("true CFG 6.0" in d["config"]["workload"]) == bool(extra)
    assert abs(d["value"] - 28 * 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]


def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run (the way the driver starts the 1-GPU
    line) must not exit with an error: it re-executes itself under torch.distributed.run (127.0.0.1 rendezvous, one process per
    GPU) and rank 0 prints the one JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--toy", "--share-gpu", "--dist-backend", "gloo",
           "--no-cpu-baseline", "--no-vanilla"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and [p["rank"] for p in d["per_rank"]] == [0, 1]


@pytest.mark.parametrize("extra", [[], ["--true-cfg", "6.0"]])
def test_bench_eight_ranks_share_the_gpu(extra):
    """BASELINE configs[3]'s launch shape without an 8-GPU node: EIGHT ranks (gloo, all on cuda:0,
    toy trunk, 256 px) under torch.distributed.run exactly as the driver launches `--gpus 8`: per-rank K_e cycling
    5 / 15 / 25 / 50 % twice, the 8-way gather of the final latents inside the timed region, the MAX-over-ranks clock, ONE
    JSON line from rank 0 with 8 `per_rank` rows and value = 28 * steps * 8 / elapsed."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 30400 + (os.getpid() % 400) + (11 if extra else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "1", "--toy", "--size", "256",
           "--share-gpu", "--dist-backend", "gloo", "--no-cpu-baseline", "--no-vanilla"] + extra
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                               # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["unit"] == "steps/s" and d["steps"] == 1
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8))
    assert [p["edit_frac"] for p in pr] == [0.05, 0.15, 0.25, 0.50] * 2
    assert all(p["edit_s"] > 0 and p["K_e"] > 0 for p in pr)
    assert [p["K_e"] for p in pr[:4]] == [p["K_e"] for p in pr[4:]] and pr[0]["K_e"] < pr[1]["K_e"] < pr[2]["K_e"] < pr[3]["K_e"]
    assert ("true CFG 6.0" in d["config"]["workload"]) == bool(extra)
    elapsed = d["ms_per_step"] * 1e-3 * d["steps"]
    assert abs(d["value"] - 28 * d["steps"] * 8 / elapsed) < 1e-6 * d["value"]
    assert elapsed >= max(p["edit_s"] for p in pr) * 0.999               # the job clock is the slowest rank's (plus the gather)


def test_cfg_branch_pairs_world_four_share_the_gpu():
    """`helper.shard_cfg_branches` at world 4 (two images, two (cond, uncond) rank pairs) with every rank on cuda:0 over gloo:
    tools/cfg_shard_run.py checks on every rank that the sharded edit is bit-identical to the unsharded one and that the two
    ranks of a pair hold the same latents (MIN-reduced over the world)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 30900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "tools/cfg_shard_run.py", "--family", "qwen", "--toy", "--size", "256", "--share-gpu",
           "--dist-backend", "gloo"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["world"] == 4 and d["shared_gpu"] and d["backend"] == "gloo"
    assert d["sharded_bit_identical_to_unsharded"] and d["pair_ranks_agree"] and 0 < d["K_e"] < d["L"] == 256


def test_bench_one_rank_executes_rccl_init_all_gather_and_max_reduce():
    """RCCL on hardware without a multi-GPU node: bench.py under torch.distributed.run with ONE rank, backend nccl (= RCCL on
    ROCm), `--force-collectives` -> process-group init on cuda:0, the barrier pair, the all_gather of the final latents inside
    the timed region and the MAX all_reduce of the elapsed time all execute; the line reports the backend it ran on."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29950 + (os.getpid() % 40)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--toy", "--dist-backend", "nccl",
           "--force-collectives", "--no-cpu-baseline", "--no-vanilla"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["collectives"] == {"backend": "nccl", "world": 1, "forced_in_world_of_one": True}
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["steps"] == 2


def test_bench_json_contract_single_gpu():
    """`python bench.py` prints exactly one JSON line with the fields the driver and SURVEY.md section 8(d) ask for (toy trunk
    so that the CPU-baseline leg takes seconds; the field set does not depend on the size)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "1", "--toy"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "steps/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    nl = d["native_library"]                     # the in-tree library, built from the kernel sources of this tree
    assert nl["path"].endswith("regione_amd/lib/libregione_hip.so") and nl["built_from_sha16"] == nl["kernel_sources_sha16"] != ""
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "traffic" in rf and "mfma_busy" in rf
    # every MFMA launch of the timed edit is on the line whichever registration carries the ops (round 4: the C++ ops bypass
    # regione_amd.ops, where the timer used to sit): toy trunk = 4 blocks -> 4 attention launches and >= 12 GEMM launches per computed step
    ra = d["roofline_attention"]
    assert ra["launches"] > 0 and ra["launches"] % 4 == 0 and rf["launches"] >= 3 * ra["launches"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "steps/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert abs(d["value"] - 28 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
