"""Output contract of the generator: (T, 4) float rows -> Standard MIDI File, with no third-party dependency.

Restates /root/reference/src/gan/utils.py:13-28,95-161 (save_piano_roll_to_midi: scale snapping, rests below the
velocity threshold, duration/step in beats, bpm clamp) and /root/reference/tools/roll_to_midi.py:1-25.  The NOTE
EVENTS are pinned against the reference function (tests/golden/midi_events.npz, tests/test_midi.py); the byte
serialisation -- SMF format 1, division 220, track 0 = tempo + 4/4 time signature, track 1 = program change +
note-on / note-on-velocity-0 pairs in pretty_midi's event order, running status -- is pinned against .mid files the
reference itself commits (good_gens1/*.mid, copied as data to tests/golden/ref_mid/): parsed to tick events and
written back, write_smf_ticks reproduces them byte for byte.
"""
from __future__ import annotations

import struct
from typing import List, Sequence, Tuple

import numpy as np

SCALES = {
    "major": [0, 2, 4, 5, 7, 9, 11], "minor": [0, 2, 3, 5, 7, 8, 10], "chromatic": list(range(12)),
    "dorian": [0, 2, 3, 5, 7, 9, 10], "phrygian": [0, 1, 3, 5, 7, 8, 10], "lydian": [0, 2, 4, 6, 7, 9, 11],
    "mixolydian": [0, 2, 4, 5, 7, 9, 10], "locrian": [0, 1, 3, 5, 6, 8, 10], "major_pentatonic": [0, 2, 4, 7, 9],
    "minor_pentatonic": [0, 3, 5, 7, 10], "blues": [0, 3, 5, 6, 7, 10],
}
NOTE_NAMES = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]
MAX_BEAT_TIME = 4.0
VELOCITY_THRESHOLD = -0.2
RESOLUTION = 220            # pretty_midi's default ticks per quarter note
GM_PROGRAMS = {"Acoustic Grand Piano": 0, "Electric Piano 1": 4, "Acoustic Guitar (nylon)": 24, "Violin": 40,
               "String Ensemble 1": 48, "Flute": 73, "Pad 2 (warm)": 89}

Note = Tuple[int, int, float, float]    # (velocity, pitch, start_sec, end_sec)


def notes_from_roll(notes_array, bpm=120.0, scale="major", root_key=0) -> Tuple[List[Note], float]:
    """utils.py:102-156.  Returns (notes, clamped bpm)."""
    bpm = max(60, min(bpm, 180))
    spb = 60.0 / bpm
    allowed = sorted((i + root_key) % 12 for i in SCALES.get(scale, SCALES["chromatic"]))

    def snap(pitch):
        octave, n = pitch // 12, pitch % 12
        return octave * 12 + min(allowed, key=lambda x: abs(x - n))

    out, t_beats = [], 0.0
    for norm_pitch, norm_velocity, norm_duration, norm_step in np.asarray(notes_array):
        step_beats = max(0.1, ((norm_step + 1.0) / 2.0) * MAX_BEAT_TIME)
        if norm_velocity < VELOCITY_THRESHOLD:
            t_beats += step_beats
            continue
        pitch = snap(int(np.clip(int((norm_pitch + 1.0) * 63.5), 36, 96)))
        vel = int(60 + ((norm_velocity - VELOCITY_THRESHOLD) / (1.0 - VELOCITY_THRESHOLD)) * 67)
        vel = int(np.clip(vel, 0, 127))
        dur_beats = max(0.25, ((norm_duration + 1.0) / 2.0) * MAX_BEAT_TIME)
        out.append((vel, int(pitch), t_beats * spb, (t_beats + dur_beats) * spb))
        t_beats += step_beats
    return out, bpm


def _vlq(n: int) -> bytes:
    b = [n & 0x7F]
    n >>= 7
    while n:
        b.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(b))


def write_smf_ticks(path: str, events: Sequence[Tuple[int, int, int]], tempo_us: int = 500000, program: int = 0) -> None:
    """Format-1 SMF from tick-level events (tick, pitch, velocity; velocity 0 = note-off), byte for byte the file
    pretty_midi 0.2.x / mido write for one instrument (pinned by tests/test_midi.py against .mid files the reference
    commits under good_gens1/): track 0 = set_tempo, 4/4 time signature, end_of_track at tick 1; track 1 = program
    change, the note events sorted by (tick, pitch * 256 + velocity) -- pretty_midi's event_compare, where a note-off is a
    note-on of velocity 0 -- written with MIDI running status, end_of_track one tick after the last event."""
    ev = sorted(events, key=lambda e: (e[0], e[1] * 256 + e[2]))
    t0 = b"\x00\xff\x51\x03" + struct.pack(">I", tempo_us)[1:] + b"\x00\xff\x58\x04\x04\x02\x18\x08"
    t0 += _vlq(1) + b"\xff\x2f\x00"
    t1 = b"\x00" + bytes([0xC0, program & 0x7F])
    cur, status = 0, None
    for tk, pitch, vel in ev:
        t1 += _vlq(tk - cur)
        if status != 0x90:
            t1 += b"\x90"
            status = 0x90
        t1 += bytes([pitch & 0x7F, vel & 0x7F])
        cur = tk
    t1 += _vlq(1) + b"\xff\x2f\x00"
    with open(path, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, 1, 2, RESOLUTION))
        for trk in (t0, t1):
            f.write(b"MTrk" + struct.pack(">I", len(trk)) + trk)


def write_smf(path: str, notes: Sequence[Note], bpm: float = 120.0, program: int = 0) -> None:
    """Notes in seconds -> ticks at `bpm` (pretty_midi's time_to_tick for a single tempo) -> write_smf_ticks."""
    tick = lambda sec: int(round(sec * bpm / 60.0 * RESOLUTION))  # noqa: E731
    ev = []
    for vel, pitch, start, end in notes:
        ev.append((tick(start), pitch, vel))
        ev.append((tick(end), pitch, 0))               # note-on with velocity 0 == note-off (pretty_midi's form)
    write_smf_ticks(path, ev, int(round(6e7 / bpm)), program)


def save_piano_roll_to_midi(notes_array, output_path, fs=100, bpm=120.0, scale="major", root_key=0,
                            instrument_name="Acoustic Grand Piano"):
    """Same signature as the reference (utils.py:95)."""
    program = GM_PROGRAMS.get(instrument_name)
    if program is None:
        print(f"[WARN] Instrument '{instrument_name}' not found. Defaulting to Piano.")
        program = 0
    notes, bpm = notes_from_roll(notes_array, bpm, scale, root_key)
    write_smf(output_path, notes, bpm, program)
    print(f"[INFO] Saved MIDI ({instrument_name} | {NOTE_NAMES[root_key]} {scale}) to {output_path}")


def roll_to_midi(roll, output_path="generated_sample.mid"):
    """tools/roll_to_midi.py:1-25: un-normalised rows [pitch, velocity, duration, start] at 120 bpm, piano."""
    notes = []
    for row in np.asarray(roll):
        start = max(0.0, float(row[3]))
        notes.append((int(max(1, min(127, row[1]))), int(np.clip(row[0], 0, 127)), start, start + max(0.05, float(row[2]))))
    write_smf(output_path, notes, 120.0, 0)
    print("Wrote", output_path)


def notes_from_ae_rows(notes_arr) -> List[Note]:
    """/root/reference/src/ae/midi_utils.py:12-35 (notes_array_to_prettymidi): rows (pitch, start, duration, velocity) in
    MIDI units / seconds; rows with pitch <= 0 or duration <= 0 are padding; pitch and velocity rounded and clipped."""
    out = []
    for p, s, d, v in np.asarray(notes_arr):
        if p <= 0 or d <= 0:
            continue
        out.append((int(np.clip(round(float(v)), 1, 127)), int(np.clip(round(float(p)), 0, 127)), float(s), float(s + d)))
    return out


def save_recon_midi(notes_in, notes_out, outdir: str, prefix: str, tempo: float = 120.0):
    """/root/reference/src/ae/midi_utils.py:37-47: <outdir>/<prefix>_in.mid and <prefix>_out.mid (the VAE trainer's
    per-epoch reconstruction dump, train_ae.py:173-188)."""
    import os
    os.makedirs(outdir, exist_ok=True)
    for arr, tag in ((notes_in, "in"), (notes_out, "out")):
        # pretty_midi refuses negative times; a (normalised) reconstruction may carry them -- clamp like roll_to_midi does
        notes = [(v, p, max(0.0, s), max(0.0, e)) for v, p, s, e in notes_from_ae_rows(arr)]
        write_smf(os.path.join(outdir, f"{prefix}_{tag}.mid"), notes, tempo, 0)


def read_smf_events(path: str):
    """Parser of the files write_smf_ticks / pretty_midi produce (format 1, one tempo, note-ons only, running status):
    returns ((format, division), tempo_us, program, [(tick, pitch, velocity)] of track 1 in file order)."""
    data = open(path, "rb").read()
    assert data[:4] == b"MThd"
    _, fmt, ntrk, div = struct.unpack(">IHHH", data[4:14])
    pos, tempo, program, events = 14, None, None, []
    for _ in range(ntrk):
        assert data[pos:pos + 4] == b"MTrk"
        ln = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        trk, p, tk, status = data[pos + 8:pos + 8 + ln], 0, 0, None
        pos += 8 + ln
        while p < len(trk):
            d = 0
            while True:
                c = trk[p]; p += 1
                d = (d << 7) | (c & 0x7F)
                if not c & 0x80:
                    break
            tk += d
            if trk[p] == 0xFF:
                mt, n = trk[p + 1], trk[p + 2]
                if mt == 0x51:
                    tempo = int.from_bytes(trk[p + 3:p + 6], "big")
                p += 3 + n
                status = None
                continue
            if trk[p] & 0x80:
                status = trk[p]
                p += 1
            if status is None:
                raise ValueError("data byte without a running status")
            if status & 0xF0 == 0xC0:
                program = trk[p]
                p += 1
            elif status & 0xF0 == 0x90:
                events.append((tk, trk[p], trk[p + 1]))
                p += 2
            else:
                raise ValueError(f"unexpected status {status:#x}")
    return (fmt, div), tempo, program, events


def read_smf_notes(path: str):
    """returns ((format, division), tempo_us, [(tick_on, tick_off, pitch, vel)]): note-ons paired first-in-first-out with
    the note-offs of their pitch."""
    hdr, tempo, _, events = read_smf_events(path)
    on, notes = {}, []
    for tk, pitch, vel in events:
        if vel:
            on.setdefault(pitch, []).append((tk, vel))
        else:
            t_on, v = on[pitch].pop(0)
            notes.append((t_on, tk, pitch, v))
    return hdr, tempo, sorted(notes)
