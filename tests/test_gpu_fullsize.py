"""BASELINE.json's configurations at FULL size through the C ABI (`pytest -m gpu`):
  cfg 3  1 000 000 mixed-length sentences (8-512 chars, unknown-word path): size-independent properties on all of them,
         oracle equality on every 16th batch of 4096;
  cfg 5  all 1 000 documents of 2048 chars against the oracle;
  cfg 4  the multi-GPU workload (corpora of seeds 100.., sentence i -> rank i mod G, chunked gather to rank 0) at ONE rank
         over RCCL: bench.py's own Workload / GpuEngine / run_job / ChunkedGather, the reassembled stream against the oracle.
Integer work: exact equality (reference src/tokenizer.rs:16-45, src/lattice.rs:101-154)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from kanpyo_amd import Tokenizer, _lib, synth

    assert _lib.lib().kgpu_device_count() > 0, "no HIP device: the gpu tests need an MI355X"
    from oracle import oracle

    oracle.build()
    sd = synth.build_dict()
    return sd, Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)


def check_properties(t, toff, status, offs):
    """What holds for every token stream whatever the corpus (the dictionary has an unknown-word entry for every category, so
    EOS is always reachable): EOS last with position == B and end == start + 3 (tokenizer.rs:27-35), tokens tile the sentence
    in bytes and in chars, ids >= 1 for words."""
    n = len(offs) - 1
    assert not status.any()
    nb = (offs[1:] - offs[:-1]).astype(np.int64)
    cnt = (toff[1:] - toff[:-1]).astype(np.int64)
    assert (cnt >= 1).all()
    last = t[toff[1:].astype(np.int64) - 1]
    assert (last["cls"] == 0).all() and (last["id"] == 0).all() and (last["byte_len"] == 0).all()
    assert np.array_equal(last["position"].astype(np.int64), nb)
    assert np.array_equal(last["end"], last["start"] + 3)
    first = np.zeros(len(t), dtype=bool)
    first[toff[:-1].astype(np.int64)] = True
    assert (t["position"][first] == 0).all() and (t["start"][first] == 0).all()
    nxt = ~first
    prev = np.nonzero(nxt)[0] - 1
    assert np.array_equal(t["position"][nxt], (t["position"] + t["byte_len"])[prev])
    assert np.array_equal(t["start"][nxt], t["end"][prev])
    words = t["cls"] != 0
    assert (t["id"][words] >= 1).all() and (t["byte_len"][words] >= 1).all()
    assert int(words.sum()) == len(t) - n  # exactly one Dummy (EOS) per sentence


def test_cfg3_one_million_sentences(full):
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 1_000_000, 2, "cfg3")
    checked = 0
    for bi, lo in enumerate(range(0, len(sents), 4096)):
        utf8, offs = pack_sentences(sents[lo : lo + 4096])
        t, toff, status = tok.tokenize_packed(utf8, offs)
        check_properties(t, toff, status, offs)
        if bi % 16 == 0:
            exp = orc.tokenize_batch(utf8, offs, 16)
            assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens), f"batch {bi}"
            checked += 1
    assert checked >= 15


def test_cfg5_all_thousand_documents(full):
    from kanpyo_amd import synth
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = full
    sents = synth.make_corpus(sd, 1000, 5, "cfg5")
    assert all(len(s) == 2048 for s in sents)
    utf8, offs = pack_sentences(sents)
    t, toff, status = tok.tokenize_packed(utf8, offs)
    check_properties(t, toff, status, offs)
    exp = orc.tokenize_batch(utf8, offs, 16)
    assert np.array_equal(toff, exp.offsets) and np.array_equal(t, exp.tokens)
    assert exp.counters["C"] == 2048 * 1000


@pytest.mark.parametrize("compact", [False, True])
def test_cfg4_workload_one_rank_over_rccl(full, compact):
    """cfg 4 at world size 1 on the RCCL backend (bench.py --force-dist): four corpora of seeds 100..103, every step's records
    gathered chunk by chunk; each gathered step, reassembled, equals the oracle's stream of the unsharded corpus.  compact: the
    8-byte records bench.py gathers by default (kgpu_tokenize_device_compact), expanded on the host by kgpu_expand_tokens."""
    import torch
    import torch.distributed as dist

    import bench
    from kanpyo_amd import synth
    from kanpyo_amd.dist import ChunkedGather, reassemble
    from kanpyo_amd.tokenizer import pack_sentences

    sd, tok, orc = full
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(36500 + os.getpid() % 2000)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        size_pg = dist.new_group(backend="gloo")
        n_per = 20_000
        corpora = [synth.make_corpus(sd, n_per, seed=100 + k, kind="cfg2") for k in range(4)]
        wl = bench.Workload(corpora, 0, 1)
        cs = bench.chunk_steps_for(wl.nb(0))
        eng = bench.GpuEngine(tok, dev, wl, queue=8, streams=4, ring=3 * cs, compact=compact)
        got = {}
        nsteps = 6  # corpora 0..3, then 0 and 1 again (ring slots reused)

        def on_chunk(c0, r):
            tok_all, cnt_all = r[0].cpu().numpy().copy(), r[1].cpu().numpy().copy()
            if compact:
                steps = range(c0, min(c0 + cs, nsteps))
                tok_all, cnt_all = bench.expand_gathered(tok_all, cnt_all, r[2], [[len(corpora[s % len(corpora)]) for s in steps]])
            got[c0] = (tok_all, cnt_all)

        bench.run_job(eng, nsteps, ChunkedGather(dst=0, size_group=size_pg), cs, on_chunk)
        eng.close()
        assert sorted(got) == list(range(0, nsteps, cs))
        want = {}
        for c0, (tok_all, cnt_all) in got.items():
            at_t = at_c = 0
            for s in range(c0, min(c0 + cs, nsteps)):
                k = s % len(corpora)
                if k not in want:
                    e = orc.tokenize_batch(*pack_sentences(corpora[k]), 16)
                    want[k] = (e.tokens.view(np.int32).reshape(-1, 6), e.offsets)
                exp_t, exp_off = want[k]
                n = len(corpora[k])
                cnt = cnt_all[at_c : at_c + n]
                nt = int(cnt.sum())
                g_tok, g_off = reassemble(tok_all[at_t : at_t + nt], cnt, n, 1)
                assert np.array_equal(g_off.astype(np.uint64), exp_off), f"step {s}"
                assert np.array_equal(g_tok, exp_t), f"step {s}"
                at_t += nt
                at_c += n
            assert at_t == len(tok_all) and at_c == len(cnt_all)
    finally:
        dist.destroy_process_group()
