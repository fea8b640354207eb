"""N > 1 path on CPU: world-size-2 gloo processes exercise the Ulysses layout/collective code of
ltxmi/distributed.py.  The layout tests replace the HIP compute by the CPU oracle's attention through the ``attn_fn``
hook; the end-to-end tests run the PRODUCT's ``usp_dit_forward`` + ``UlyssesAttnProcessor`` + ``Transformer3DModel``
with ``ltxmi.ops`` swapped for its CPU double (tests/cpu_ops_double.py: same argument meaning as the kernels, incl. the
destination-major pack, the segmented attention output and the K-blocked GEMM operand the zero-copy exchange uses)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, q):
    for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn_name](rank, world)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, res in results:
        assert res == "ok", f"rank {rank}:\n{res}"


def _global_qkv(B=2, N=12, H=4, dh=8):
    g = torch.Generator().manual_seed(0)
    return torch.randn(B, N, 3, H, dh, generator=g)


def _case_layout(rank, world):
    from ltxmi import distributed as sp
    qkv = _global_qkv()
    B, N, _, H, dh = qkv.shape
    local = sp.shard_tokens(qkv, rank, world).contiguous()
    assert local.shape == (B, N // world, 3, H, dh)
    full = sp.seq_to_head_shard(local)
    Hl = H // world
    assert torch.equal(full, qkv[:, :, :, rank * Hl:(rank + 1) * Hl])          # all tokens, my heads
    back = sp.head_to_seq_shard(full[:, :, 0].contiguous())                     # q: all heads, my tokens
    assert torch.equal(back, local[:, :, 0])
    gathered = sp.gather_tokens(local)
    assert torch.equal(gathered, qkv)
    with pytest.raises(ValueError):
        sp.shard_tokens(qkv[:, :11], rank, world)
    with pytest.raises(ValueError):
        sp.seq_to_head_shard(local[:, :, :, :3].contiguous())


def _case_attention(rank, world):
    from ltxmi import distributed as sp
    from oracle import dit
    qkv = _global_qkv(B=3, N=16, H=4, dh=16)
    ref = dit.sdpa_nhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])               # full attention

    def attn_fn(q, k, v, scale):
        assert abs(scale - 0.25) < 1e-9
        return dit.sdpa_nhd(q, k, v)

    local = sp.shard_tokens(qkv, rank, world).contiguous()
    out = sp.usp_attn_forward(local, 0.25, attn_fn=attn_fn)
    torch.testing.assert_close(out, sp.shard_tokens(ref, rank, world), rtol=1e-5, atol=1e-6)


def _case_clock(rank, world):
    """bench.py's N > 1 protocol: barrier, MAX all-reduce of the elapsed time."""
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)


def _dit_setup(per_token):
    """A tiny bf16 DiT of the product on the CPU (ops double installed) with full-size inputs, identical on every rank."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_double
    cpu_ops_double.install()
    import ltxmi
    from oracle import dit, sched
    bf = torch.bfloat16
    cfg = dict(dit.default_2b_config(), num_attention_heads=4, attention_head_dim=64, num_layers=3,
               cross_attention_dim=256, caption_channels=128)
    sd32 = {k: v.to(bf).float() for k, v in dit.init_state_dict(cfg, seed=3).items()}
    f, h, w = 2, 3, 4
    N, B, T = f * h * w, 3, 12
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, N, 128, generator=g).to(bf)
    enc = torch.randn(B, T, 128, generator=g).to(bf)
    mask = torch.ones(B, T)
    mask[:, 8:] = 0
    if per_token:
        ts = torch.full((B, N), 0.7)
        ts[:, : h * w] = 0.1                               # a conditioned first latent frame
    else:
        ts = torch.full((B, 1), 0.7)
    m = ltxmi.Transformer3DModel(**cfg)
    m.load_state_dict(sd32)
    m = m.to(bf).eval()
    fc = m.precompute_freqs_cis(sched.fractional_coords(f, h, w, 1, 25.0))
    skip = m.create_skip_layer_mask(1, 3, 2, [1])
    kw = dict(encoder_hidden_states=enc, encoder_attention_mask=mask, timestep=ts, skip_layer_mask=skip,
              skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, latent_shape=(f, h, w))
    return ltxmi, m, x, fc, kw


def _usp_case(rank, world, per_token):
    from ltxmi import distributed as sp
    ltxmi, m, x, fc, kw = _dit_setup(per_token)
    with torch.no_grad():
        ref = m(x.clone(), freqs_cis=fc, return_dict=False, **kw)[0]                 # one rank, default processor
        sp.enable_sequence_parallel(m, overlap=False)
        plain = sp.usp_dit_forward(m, x.clone(), fc, **kw)[0]                        # tokens sharded over the ranks
        sp.enable_sequence_parallel(m)                                               # default: two micro-batches of rows
        assert m._sp_overlap
        out = sp.usp_dit_forward(m, x.clone(), fc, **kw)[0]
    assert out.shape == ref.shape
    # the micro-batched block loop (what hides the exchanges behind the other rows' kernels) computes every row as the
    # plain loop does
    assert torch.equal(out, plain)
    err = float((out.float() - ref.float()).norm() / ref.float().norm())
    # every op is row-wise: the two runs differ only by matmul blocking (row counts differ) before a bf16 rounding
    assert err < 2e-3, err
    # and the perturbed (STG) row really differs from the text row
    assert float((out[2].float() - out[1].float()).norm()) > 1e-3


def _case_usp_dit_forward(rank, world):
    _usp_case(rank, world, per_token=False)


def _case_usp_dit_forward_per_token(rank, world):
    _usp_case(rank, world, per_token=True)


def _case_usp_interrupt_is_collective(rank, world):
    """ltxv_model._interrupt raised on ONE rank: the ranks agree on it without a host synchronisation in the step (the
    flag posted by forward k is read by forward k + 1), so the forward in which it was raised still completes on EVERY
    rank and the next one returns [None] on EVERY rank -- nobody is left waiting in an all-to-all."""
    from ltxmi import distributed as sp
    ltxmi, m, x, fc, kw = _dit_setup(False)
    sp.enable_sequence_parallel(m)

    class Holder:
        _interrupt = (rank == 1)

    class Quiet:
        _interrupt = False

    with torch.no_grad():
        out = sp.usp_dit_forward(m, x.clone(), fc, ltxv_model=Quiet(), **kw)
        assert out[0] is not None and out[0].shape == (3, 24, 128)
        first = sp.usp_dit_forward(m, x.clone(), fc, ltxv_model=Holder(), **kw)       # raised on rank 1 only: posted
        assert first[0] is not None and torch.equal(first[0], out[0])
        assert sp.usp_dit_forward(m, x.clone(), fc, ltxv_model=Quiet(), **kw) == [None]   # agreed: every rank leaves
        again = sp.usp_dit_forward(m, x.clone(), fc, ltxv_model=Quiet(), **kw)        # consumed: back to normal
    assert again[0] is not None and torch.equal(again[0], out[0])


def _case_usp_first_forward_of_a_fresh_model(rank, world):
    """the overlap mode (two micro-batches of rows, a stream each) as the VERY FIRST forward of a freshly
    loaded model, and again after the packed weights were invalidated (reload / LoRA merge): everything the slices share
    is built before the streams fork (BasicTransformerBlock.prepare_shared_state), so the result is the plain loop's.
    (On the CPU the slices run one after the other -- the GPU twin of this case is in tests/test_gpu_model.py.)  With
    step-invariant caching off, the stacked text K/V of the forward serve EVERY slice: no per-layer K/V GEMM is left."""
    from ltxmi import distributed as sp
    ltxmi, m, x, fc, kw = _dit_setup(False)
    _, m2, _, _, _ = _dit_setup(False)
    with torch.no_grad():
        sp.enable_sequence_parallel(m)                                               # overlap on: the first forward ever
        out = sp.usp_dit_forward(m, x.clone(), fc, **kw)[0]
        sp.enable_sequence_parallel(m2, overlap=False)
        plain = sp.usp_dit_forward(m2, x.clone(), fc, **kw)[0]
        assert torch.equal(out, plain)
        for blk in m.transformer_blocks:
            blk.attn1.invalidate_packed()
            blk.attn2.invalidate_packed()
        assert torch.equal(sp.usp_dit_forward(m, x.clone(), fc, **kw)[0], plain)
        # caching off: the stacked projection of all layers' text K/V, handed to the blocks per slice
        from ltxmi import ops
        ltxmi.set_step_invariant_caching(False)
        try:
            D = m.inner_dim
            per_layer = []
            real = ops.gemm

            def counting(a, w, *args, **kws):
                if tuple(w.shape) == (2 * D, D):
                    per_layer.append(tuple(a.shape))
                return real(a, w, *args, **kws)

            ops.gemm = counting
            off = sp.usp_dit_forward(m, x.clone(), fc, **kw)[0]
            ops.gemm = real
        finally:
            ltxmi.set_step_invariant_caching(True)
        assert torch.equal(off, plain)
        assert per_layer == [], per_layer


def _case_usp_interrupt_on_the_last_forward(rank, world):
    """an interrupt raised during the LAST forward of a generation is posted but never read (there is no
    forward k + 1).  ``begin_generation`` drops it, so the next generation's first forward runs normally on every rank."""
    from ltxmi import distributed as sp
    ltxmi, m, x, fc, kw = _dit_setup(False)
    sp.enable_sequence_parallel(m)

    class Holder:
        _interrupt = (rank == 1)

    class Quiet:
        _interrupt = False

    with torch.no_grad():
        out = sp.usp_dit_forward(m, x.clone(), fc, ltxv_model=Quiet(), **kw)
        last = sp.usp_dit_forward(m, x.clone(), fc, ltxv_model=Holder(), **kw)        # raised in the generation's last forward
        assert last[0] is not None
        sp.begin_generation(m)                                                         # (the pipeline calls this)
        new = sp.usp_dit_forward(m, x.clone(), fc, ltxv_model=Quiet(), **kw)
    assert new[0] is not None and torch.equal(new[0], out[0])


def _case_pipeline_runs_sequence_parallel_unchanged(rank, world):
    """``enable_sequence_parallel(model, bind_forward=True)`` (the reference's MethodType pattern, wan/text2video.py):
    ``ltxmi.LTXVideoPipeline.__call__`` -- the reference's signature -- then denoises with the tokens sharded over the
    ranks without knowing it: same latents as the single-rank loop up to matmul blocking, identical on every rank."""
    from ltxmi import distributed as sp
    ltxmi, m, x, fc, kw = _dit_setup(False)
    g = torch.Generator().manual_seed(9)
    T = 12
    pos, neg = torch.randn(1, T, 128, generator=g), torch.randn(1, T, 128, generator=g)
    pmask, nmask = torch.ones(1, T), torch.ones(1, T)
    pmask[:, 8:] = 0
    nmask[:, 3:] = 0
    noise = torch.randn(1, 2 * 2 * 4, 128, generator=g)
    args = dict(height=64, width=128, num_frames=9, frame_rate=25.0, prompt_embeds=pos, prompt_attention_mask=pmask,
                negative_prompt_embeds=neg, negative_prompt_attention_mask=nmask, num_inference_steps=2, guidance_scale=3.0,
                stg_scale=1.0, rescaling_scale=0.7, skip_block_list=[1],
                skip_layer_strategy=ltxmi.SkipLayerStrategy.AttentionValues, latents=noise, output_type="latent",
                is_video=True, joint_pass=True, return_dict=False)
    pipe = ltxmi.LTXVideoPipeline(transformer=m, scheduler=ltxmi.RectifiedFlowScheduler(shifting="SD3", target_shift_terminal=0.1))
    single = pipe(**args)[0]
    sp.enable_sequence_parallel(m, bind_forward=True)
    seen = []
    sharded = pipe(callback=lambda i, lat, start, **k: seen.append(i), **args)[0]
    assert seen == [-1, 0, 1]
    assert sharded.shape == single.shape == (1, 128, 2, 2, 4)
    err = float((sharded.float() - single.float()).norm() / single.float().norm())
    assert err < 5e-3, err
    both = [torch.empty_like(sharded) for _ in range(world)]
    dist.all_gather(both, sharded.contiguous())
    assert all(torch.equal(b, both[0]) for b in both)
    # a cancel raised on one rank: agreed one step later, and the pipeline returns None on EVERY rank
    holder = type("H", (), {"_interrupt": rank == 0})()
    assert pipe(ltxv_model=holder, **dict(args, num_inference_steps=3)) is None
    # ... and the next generation is not affected by the flag the cancelled one posted
    again = pipe(**args)[0]
    assert torch.equal(again, sharded)
    sp.disable_sequence_parallel(m)
    assert torch.equal(pipe(**args)[0], single)


def _case_exchange_tiles(rank, world):
    """The two-phase tile exchange of tile_parallel_vae_decode: every rank decodes ALL its tiles (n = rank mod P) before
    the one collective; uneven tile counts and sizes; the tiles come back as they were written."""
    from ltxmi import distributed as sp
    shapes = [(1, 3, 33, 4, 6)] + [(1, 3, 32, 4, 6)] * 3 + [(1, 3, 9, 4, 6)]        # 5 tiles: rank 0 owns 3, rank 1 owns 2
    want = [torch.randn(s, generator=torch.Generator().manual_seed(100 + n)).to(torch.float16) for n, s in enumerate(shapes)]
    decoded = []

    def make(n):
        def dec(out):
            assert tuple(out.shape) == shapes[n] and out.is_contiguous() and out.dtype == torch.float16
            decoded.append(n)
            out.copy_(want[n])
        return dec

    trace = []
    tiles = sp.exchange_tiles([make(n) for n in range(5)], shapes, dtype=torch.float16, trace=trace)
    assert decoded == list(range(rank, 5, world))                                     # only its own tiles
    assert len(tiles) == 5 and all(torch.equal(t, w) and t.is_contiguous() for t, w in zip(tiles, want))
    kinds = [k for k, _ in trace]
    assert kinds.count("collective") == 1 and kinds.index("collective") == len(decoded) == len(kinds) - 1, trace
    # in-place edits of one tile (the blends) must not touch its neighbours in the receive buffer
    tiles[1].zero_()
    assert torch.equal(tiles[0], want[0]) and torch.equal(tiles[2], want[2]) and torch.equal(tiles[3], want[3])


def test_ulysses_layout_world2():
    _run("_case_layout")


def test_ulysses_attention_equals_full_attention_world2():
    _run("_case_attention")


def test_bench_clock_protocol_world2():
    _run("_case_clock")


def test_usp_dit_forward_world2():
    _run("_case_usp_dit_forward")


def test_usp_dit_forward_per_token_timesteps_world2():
    _run("_case_usp_dit_forward_per_token")


def test_usp_interrupt_is_collective_world2():
    _run("_case_usp_interrupt_is_collective")


def test_exchange_tiles_two_phase_world2():
    _run("_case_exchange_tiles")


def test_exchange_tiles_world3():
    _run("_case_exchange_tiles", world=3)


def test_usp_first_forward_of_a_fresh_model_world2():
    _run("_case_usp_first_forward_of_a_fresh_model")


def test_usp_interrupt_on_the_last_forward_world2():
    _run("_case_usp_interrupt_on_the_last_forward")


def test_pipeline_runs_sequence_parallel_unchanged_world2():
    _run("_case_pipeline_runs_sequence_parallel_unchanged")
