"""N > 1 path on CPU: world-size-2 gloo processes exercise the Ulysses layout/collective code of
ltxmi/distributed.py (the HIP compute is replaced by the CPU oracle's attention through the
``attn_fn`` hook, which is the only thing that differs from the GPU run)."""
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
    results = [q.get(timeout=120) for _ in procs]
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


def test_ulysses_layout_world2():
    _run("_case_layout")


def test_ulysses_attention_equals_full_attention_world2():
    _run("_case_attention")


def test_bench_clock_protocol_world2():
    _run("_case_clock")
