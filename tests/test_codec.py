"""Token <-> string codec and checkpoint loading (CPU)."""
import numpy as np
import pytest
import torch

import drakes_amd
from drakes_amd import codec


def test_round_trip_and_reference_table():
    g = np.random.Generator(np.random.PCG64(1))
    toks = g.integers(0, 4, (7, 200))
    strs = codec.batch_dna_detokenize(toks)
    assert len(strs) == 7 and all(len(s) == 200 and set(s) <= set('ACGT') for s in strs)
    assert np.array_equal(codec.batch_dna_tokenize(strs), toks)
    assert codec.dna_detokenize([0, 1, 2, 3, 3]) == 'ACGTT' and codec.dna_tokenize('TGCA') == [3, 2, 1, 0]
    assert codec.batch_dna_detokenize(torch.from_numpy(toks))[3] == strs[3]
    assert codec.batch_dna_tokenize([]).shape == (0, 0)


def test_errors():
    with pytest.raises(KeyError):
        codec.batch_dna_tokenize(['ACGN'])
    with pytest.raises(KeyError):
        codec.dna_tokenize('acgt')
    with pytest.raises(ValueError):
        codec.batch_dna_tokenize(['ACG', 'AC'])
    with pytest.raises(IndexError):
        codec.batch_dna_detokenize(np.array([[0, 4]]))       # mask token: no letter


def test_checkpoint_forms(tmp_path, det_sd):
    bare, lightning = tmp_path / 'bare.ckpt', tmp_path / 'pl.ckpt'
    torch.save(det_sd, bare)
    torch.save({'state_dict': dict(det_sd, **{'noise.dummy': torch.zeros(1)}), 'epoch': 3, 'global_step': 10}, lightning)
    for p in (bare, lightning):
        sd = codec.load_checkpoint_state_dict(str(p))
        assert sd.keys() == det_sd.keys()
        m = drakes_amd.Diffusion(drakes_amd.default_config())
        m.load_state_dict(sd, strict=True)
        assert torch.equal(m.state_dict()['backbone.convs.7.weight'], det_sd['backbone.convs.7.weight'])
    torch.save({'foo': 1}, tmp_path / 'bad.ckpt')
    with pytest.raises(ValueError):
        codec.load_checkpoint_state_dict(str(tmp_path / 'bad.ckpt'))
