"""CPU: the generator's output contract -- note events pinned against the reference's save_piano_roll_to_midi
(fixture recorded by tests/golden/make_golden.py::midi_case), SMF structure checked by parsing it back."""
import os

import numpy as np

import melo_gan_amd  # noqa: F401
from melo_gan_amd import midi

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "midi_events.npz"))
CASES = {"a": dict(bpm=120.0, scale="major", root_key=0),
         "b": dict(bpm=200.0, scale="minor_pentatonic", root_key=7),
         "c": dict(bpm=45.0, scale="not_a_scale", root_key=3)}


def test_note_events_match_reference():
    for tag, kw in CASES.items():
        notes, bpm = midi.notes_from_roll(G["roll"], **kw)
        ref = G[f"{tag}.notes"]
        assert bpm == float(G[f"{tag}.tempo"])
        assert len(notes) == len(ref)
        got = np.array(notes, dtype=np.float64)
        assert np.array_equal(got[:, :2], ref[:, :2])                      # velocity, pitch: exact integers
        np.testing.assert_allclose(got[:, 2:], ref[:, 2:], rtol=0, atol=1e-9)


def test_smf_roundtrip(tmp_path):
    p = str(tmp_path / "x.mid")
    midi.save_piano_roll_to_midi(G["roll"], p, bpm=200.0, scale="minor_pentatonic", root_key=7, instrument_name="Violin")
    (fmt, div), tempo, notes = midi.read_smf_notes(p)
    assert fmt == 1 and div == 220 and tempo == int(round(6e7 / 180))      # bpm clamped to 180
    ref = G["b.notes"]
    assert len(notes) == len(ref)
    tick = lambda s: int(round(s * 180 / 60 * 220))  # noqa: E731
    # overlapping notes of one pitch make on/off PAIRING ambiguous in any SMF: compare the event multisets
    assert sorted((n[0], n[2], n[3]) for n in notes) == sorted((tick(s), int(pi), int(v)) for v, pi, s, e in ref)
    assert sorted((n[1], n[2]) for n in notes) == sorted((tick(e), int(pi)) for v, pi, s, e in ref)
    midi.roll_to_midi(np.array([[60, 100, 0.5, 0.0], [200, 0, 0.01, -1.0]]), str(tmp_path / "y.mid"))
    _, _, n2 = midi.read_smf_notes(str(tmp_path / "y.mid"))
    assert n2 == [(0, 22, 127, 1), (0, 220, 60, 100)]
