"""CPU: the generator's output contract -- note events pinned against the reference's save_piano_roll_to_midi
(fixture recorded by tests/golden/make_golden.py::midi_case); the SMF byte serialisation pinned against .mid files the
reference commits (tests/golden/ref_mid/: copies of good_gens1/test_happy_1.mid, good_gens1/test_sad_2.mid and
generated_tests/test_calm_2.mid -- pretty_midi output, data not source)."""
import glob
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


def test_smf_bytes_match_the_reference_committed_files(tmp_path):
    """Parse a file pretty_midi wrote for the reference into tick events, write them back: identical bytes (header,
    tempo / time-signature track, program change, event order at equal ticks, running status, end-of-track deltas)."""
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_mid", "*.mid")))
    assert len(files) == 3
    for f in files:
        (fmt, div), tempo, program, events = midi.read_smf_events(f)
        assert (fmt, div) == (1, midi.RESOLUTION) and len(events) >= 998
        out = str(tmp_path / os.path.basename(f))
        midi.write_smf_ticks(out, list(reversed(events)), tempo, program)     # order is the writer's business
        assert open(out, "rb").read() == open(f, "rb").read(), f
    # and the seconds -> ticks path: notes rebuilt from the parsed ticks at the file's tempo give the same bytes
    (fmt, div), tempo, program, events = midi.read_smf_events(files[0])
    _, _, notes = midi.read_smf_notes(files[0])
    bpm = 6e7 / tempo
    sec = lambda tk: tk * 60.0 / (bpm * div)  # noqa: E731
    out = str(tmp_path / "from_seconds.mid")
    midi.write_smf(out, [(v, p, sec(a), sec(b)) for a, b, p, v in notes], bpm, program)
    assert open(out, "rb").read() == open(files[0], "rb").read()


def test_ae_reconstruction_midi(tmp_path):
    """save_recon_midi (src/ae/midi_utils.py:12-47): rows (pitch, start, duration, velocity); padding rows (pitch <= 0 or
    duration <= 0) skipped; pitch / velocity rounded and clipped to [0,127] / [1,127]."""
    rows = np.array([[60.4, 0.0, 0.5, 100.6], [0.0, 1.0, 0.5, 90.0], [72.0, 1.0, 0.0, 90.0], [130.0, 2.0, 1.0, 0.2]], np.float32)
    assert midi.notes_from_ae_rows(rows) == [(101, 60, 0.0, 0.5), (1, 127, 2.0, 3.0)]
    midi.save_recon_midi(rows, rows[:1], str(tmp_path), "ep1_x")
    _, tempo, n_in = midi.read_smf_notes(str(tmp_path / "ep1_x_in.mid"))
    _, _, n_out = midi.read_smf_notes(str(tmp_path / "ep1_x_out.mid"))
    assert tempo == 500000 and n_in == [(0, 220, 60, 101), (880, 1320, 127, 1)] and n_out == [(0, 220, 60, 101)]
