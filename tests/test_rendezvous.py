"""CPU: the torch-free control plane of the one-process-per-GPU runs (evidence_amd/rendezvous.py) at world sizes 2,
3 and 8 — the exchange bench.py and the sharded samplers use beside RCCL: id broadcast, barrier, max / min over ranks,
all-gather of small host buffers — and the socket transport of the sharded log-L."""
import multiprocessing as mp
import time
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


def _worker(rank, world, address, q):
    sys.path.insert(0, str(REPO))
    from evidence_amd.rendezvous import Rendezvous
    from evidence_amd.sharded import ShardedLogLike, ShardedWalker, partition
    with Rendezvous(rank, world, address=address, timeout=30) as rz:
        uid = rz.broadcast(bytes(range(128)) if rank == 0 else None, src=0)
        got = rz.allgather({"rank": rank, "x": np.full(3, float(rank))})
        rz.barrier()
        mx, mn, sm = rz.allreduce(1.5 * rank, "max"), rz.allreduce(rank + 7, "min"), rz.allreduce(rank, "sum")
        theta = np.random.default_rng(1).random((41, 4))
        calls = []
        def evaluate(x):
            calls.append(len(x))
            return -(x ** 2).sum(axis=1)
        ll = ShardedLogLike(rank, world, evaluate=evaluate, transport="rdzv", group=rz)(theta)
        def walk(cube, theta, logl, lstar, chol, wrapped, nsteps, max_rounds, seed, walker_base=0):
            return cube + walker_base, theta, logl - seed, len(cube)
        wc, wt, wl, used = ShardedWalker(rank, world, walk, transport="rdzv", group=rz)(
            np.zeros((10, 2)), np.ones((10, 2)), np.arange(10.0), 0.0, np.eye(2), None, 3, 9, 5)
        lo, hi = partition(41, world)[rank]
        q.put((rank, uid == bytes(range(128)), [g["rank"] for g in got], float(got[-1]["x"][0]), mx, mn, sm,
               bool(np.array_equal(ll, -(theta ** 2).sum(axis=1))), calls == [hi - lo],
               wc[:, 0].tolist(), float(wl[3]), used))


@pytest.mark.parametrize("world", [2, 3, 8])        # 8: the node size the driver launches; 41 and 10 rows: ragged shards
def test_rendezvous_collectives_and_socket_transport(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    address = f"unix:rvll-test-{os.getpid()}-{world}"
    procs = [ctx.Process(target=_worker, args=(r, world, address, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    from evidence_amd.sharded import partition
    base = np.concatenate([np.full(hi - lo, float(lo)) for lo, hi in partition(10, world)]).tolist()
    for rank, uid_ok, ranks, lastx, mx, mn, sm, ll_ok, calls_ok, wc0, wl3, used in results:
        assert uid_ok and ranks == list(range(world)) and lastx == float(world - 1)
        assert mx == 1.5 * (world - 1) and mn == 7 and sm == sum(range(world))
        assert ll_ok and calls_ok
        assert wc0 == base and wl3 == 3.0 - 5 and used == 10      # every shard saw the common seed and its own base


def test_rendezvous_over_tcp_and_single_rank():
    from evidence_amd.rendezvous import Rendezvous, default_address
    with Rendezvous(0, 1) as rz:                                   # world 1: no socket at all
        assert rz.allgather(5) == [5] and rz.allreduce(2.0, "max") == 2.0 and rz.broadcast("a") == "a"
    assert default_address({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29512", "TORCHELASTIC_RUN_ID": "r7"}) == \
        "unix:rvll-rdzv-127.0.0.1-29512-r7"
    assert default_address({"RVLL_RDZV": "tcp://127.0.0.1:1234"}) == "tcp://127.0.0.1:1234"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from evidence_amd.rendezvous import Rendezvous\n"
            "r = Rendezvous(int(sys.argv[1]), 2, address=sys.argv[2], timeout=30)\n"
            "print(r.allreduce(int(sys.argv[1]) + 1, 'sum')); r.close()\n") % str(REPO)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RVLL_RDZV_SECRET="s3cret-of-this-test")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), f"tcp://127.0.0.1:{port}"], stdout=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=60)[0].strip() for p in procs]
    assert outs == ["3", "3"] and all(p.returncode == 0 for p in procs)
    # over tcp there is no uid to check: without a secret of its own the rendezvous refuses (the run id is not one — ADVICE r3)
    from evidence_amd.rendezvous import RendezvousError, shared_key
    with pytest.raises(RendezvousError, match="RVLL_RDZV_SECRET"):
        shared_key(f"tcp://127.0.0.1:{port}", env={"TORCHELASTIC_RUN_ID": "none"})
    assert shared_key("unix:x", env={"TORCHELASTIC_RUN_ID": "none"}) != shared_key("unix:x", env={"RVLL_RDZV_SECRET": "k"})


def test_the_control_plane_and_the_bench_launch_path_do_not_import_torch():
    """A process that imports torch first binds torch's bundled HIP runtime and RCCL: neither the
    package, nor the rendezvous, nor bench.py may pull it in."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import evidence_amd, evidence_amd.rendezvous, evidence_amd.sharded, evidence_amd.nested, evidence_amd.callbacks\n"
            "import importlib.util\n"
            "spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "print('torch' in sys.modules)\n") % (str(REPO), str(REPO / "bench.py"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "False"


def test_wire_format_carries_a_closed_set_of_values_and_nothing_else():
    """the first version unpickled whatever arrived.  The codec now carries plain values only; decoding
    never constructs anything but numbers, strings, bytes, arrays and containers of those."""
    from evidence_amd.rendezvous import RendezvousError, decode, encode
    value = {"rank": 3, "x": np.arange(6, dtype=np.float64).reshape(2, 3), "id": bytes(range(128)), "ok": True,
             "none": None, "big": 2 ** 80, "t": (1.5, "s", [np.int32(7), np.float32(0.25)]),
             "flags": np.array([1, 0, 2], dtype=np.int32), "empty": np.empty((0, 4))}
    back = decode(encode(value))
    assert back["rank"] == 3 and back["id"] == bytes(range(128)) and back["ok"] is True and back["none"] is None
    assert back["big"] == 2 ** 80 and back["t"] == (1.5, "s", [7, 0.25])
    assert np.array_equal(back["x"], value["x"]) and back["x"].dtype == np.float64 and back["x"].flags.writeable
    assert np.array_equal(back["flags"], value["flags"]) and back["flags"].dtype == np.int32
    assert back["empty"].shape == (0, 4)
    for bad in (object(), {1, 2}, np.array(["a"]), np.array([1 + 2j]), lambda: 0):
        with pytest.raises(RendezvousError):
            encode(bad)
    import pickle
    for junk in (pickle.dumps({"a": 1}), b"", b"a\xff\x01", encode([1, 2]) + b"x", b"l" + (2 ** 40).to_bytes(8, "big"),
                 b"a\x00\x01" + (2 ** 50).to_bytes(8, "big")):
        with pytest.raises(RendezvousError):
            decode(junk)


def _auth_worker(rank, address, key, q):
    sys.path.insert(0, str(REPO))
    from evidence_amd.rendezvous import Rendezvous, RendezvousError
    try:
        with Rendezvous(rank, 2, address=address, timeout=6, key=key) as rz:
            q.put((rank, rz.allreduce(rank + 1, "sum")))
    except RendezvousError as exc:
        q.put((rank, f"error: {exc}"))


def test_a_peer_without_the_key_is_not_admitted_and_a_tampered_frame_is_refused():
    import socket
    import struct
    from evidence_amd.rendezvous import Rendezvous, RendezvousError, _Channel, _recv_exact, encode
    ctx = mp.get_context("spawn")
    # 1. a rank with another key never joins: rank 0 keeps waiting and reports who is missing; the stranger sees the
    #    connection dropped
    q = ctx.Queue()
    address = f"unix:rvll-auth-{os.getpid()}"
    procs = [ctx.Process(target=_auth_worker, args=(0, address, b"k" * 32, q)),
             ctx.Process(target=_auth_worker, args=(1, address, b"x" * 32, q))]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
    assert "only 1 of 2 ranks arrived" in got[0] and str(got[1]).startswith("error")
    # 2. a pickle (what the first version would have unpickled) sent instead of the fixed-size hello is dropped, and the
    #    genuine rank that connects afterwards is admitted
    address = f"unix:rvll-auth2-{os.getpid()}"
    p0 = ctx.Process(target=_auth_worker, args=(0, address, b"k" * 32, q))
    p0.start()
    import pickle
    deadline = time.monotonic() + 30
    while True:
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            s.connect("\0" + address.partition(":")[2])
            break
        except OSError:
            s.close()
            assert time.monotonic() < deadline
            time.sleep(0.05)
    data = pickle.dumps(1)
    s.sendall(struct.pack("!Q", len(data)) + data + b"\0" * 64)
    p1 = ctx.Process(target=_auth_worker, args=(1, address, b"k" * 32, q))
    p1.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    s.close()
    for p in (p0, p1):
        p.join(timeout=30)
    assert got == {0: 3, 1: 3}
    # 3. frames: a flipped payload bit, a replayed frame and a frame from the wrong direction all fail the MAC
    a, b = socket.socketpair()
    tx, rx = _Channel(a, b"k" * 32, 1, 0), _Channel(b, b"k" * 32, 0, 1)
    tx.send({"v": np.arange(4.0)})
    assert np.array_equal(rx.recv()["v"], np.arange(4.0))
    payload = encode([1, 2, 3])
    frame = struct.pack("!Q", len(payload)) + tx._mac(tx.tx_tag, tx.tx, payload) + payload
    a.sendall(frame[:-1] + bytes([frame[-1] ^ 1]))
    with pytest.raises(RendezvousError, match="authentication"):
        rx.recv()
    a.sendall(frame)                                   # the genuine frame: accepted once ...
    assert rx.recv() == [1, 2, 3]
    a.sendall(frame)                                   # ... and refused when replayed (the counter moved on)
    with pytest.raises(RendezvousError, match="authentication"):
        rx.recv()
    a.close(); b.close()
