"""The windowed kernel's TEAM form (kgpu_window.hip: two wavefronts per sentence, window k on wavefront k mod 2, a structure token and a value token passed
from window to window) against the oracle, forced on for everything: KGPU_POOL=0 makes every chain start with the windowed kernel, KGPU_WINDOW_TEAM=2 makes that
its team form whatever the load.  What the form hands on (a carry list beyond its banks) comes back through the ordinary form behind it.  The reference builds
positions independently and relaxes them in order (src/lattice.rs:101-114, 116-142): which wavefront ran which window cannot show in the records."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    from kanpyo_amd import _lib

    assert _lib.lib().kgpu_device_count() > 0, "no HIP device: the gpu tests need an MI355X"
    from oracle import oracle

    oracle.build()
    return _lib, oracle


def _same(tok, orc, sentences, what):
    from kanpyo_amd.tokenizer import pack_sentences

    utf8, offs = pack_sentences(sentences)
    exp = orc.tokenize_batch(utf8, offs, 16)
    got_t, got_off, status = tok.tokenize_packed(utf8, offs)
    assert not status.any(), what
    assert np.array_equal(got_off, exp.offsets), f"{what}: per-sentence token counts differ"
    assert np.array_equal(got_t, exp.tokens), f"{what}: records differ"


@pytest.mark.parametrize("window_kib", ["10", "9", "16"])
def test_team_form_everything_through_it(libs, window_kib, monkeypatch):
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    monkeypatch.setenv("KGPU_POOL", "0")
    monkeypatch.setenv("KGPU_WINDOW", window_kib)
    monkeypatch.setenv("KGPU_WINDOW_TEAM", "2")
    sd = synth.build_dict(20000, seed=11)
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    _same(tok, orc, synth.EDGE_SENTENCES + ["テ", "テあ", "1" * 1024, "1" * 1025, "ア" * 1500, "漢字かな" * 300], "edge sentences")
    _same(tok, orc, synth.make_corpus(sd, 60, 5, "cfg5"), "cfg 5 documents")
    _same(tok, orc, synth.make_corpus(sd, 1500, 2, "cfg3"), "cfg 3 mix")
    _same(tok, orc, synth.make_corpus(sd, 2000, 1, "cfg2"), "cfg 2")
    rng = random.Random(int(window_kib))
    for k in range(4):
        d, sents = synth.dense_case(rng) if k % 2 == 0 else synth.width_case(rng)
        t2 = Tokenizer(d)
        _same(t2, oracle.OracleTokenizer.from_dict(d), sents, f"fuzz dictionary {k}")
        t2.close()
        _same(tok, orc, synth.mixed_case(sd, rng, sizes=(1, 5, 50, 700)), f"mixed batch {k}")
    r = tok.routing()
    assert r["window_reruns"] == 0
    tok.close()


def test_lone_batch_of_documents_takes_the_team_form_by_itself(libs):
    """Defaults: a lone window-first batch (1000 documents against 4096 wavefront slots) is what the team form is for; the 392k-record dictionary of the configs."""
    from kanpyo_amd import Tokenizer, synth

    _, oracle = libs
    sd = synth.build_dict()
    tok, orc = Tokenizer(sd.dict), oracle.OracleTokenizer.from_dict(sd.dict)
    docs = synth.make_corpus(sd, 300, 5, "cfg5")
    for _ in range(2):
        _same(tok, orc, docs, "cfg 5 documents, defaults")
    tok.close()
