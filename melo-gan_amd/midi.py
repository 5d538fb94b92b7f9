"""Output contract of the generator: (T, 4) float rows -> Standard MIDI File, with no third-party dependency.

Restates /root/reference/src/gan/utils.py:13-28,95-161 (save_piano_roll_to_midi: scale snapping, rests below the
velocity threshold, duration/step in beats, bpm clamp) and /root/reference/tools/roll_to_midi.py:1-25.  The NOTE
EVENTS are pinned against the reference function (tests/golden/midi_events.npz, tests/test_midi.py); the byte
serialisation follows pretty_midi's documented layout -- SMF format 1, division 220, track 0 = tempo + 4/4 time
signature, track 1 = program change + note-on / note-on-velocity-0 pairs -- but is NOT pinned byte-for-byte
(pretty_midi is not installable here): parity unpinned for the serialisation only.
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


def write_smf(path: str, notes: Sequence[Note], bpm: float = 120.0, program: int = 0) -> None:
    """Format-1 SMF: meta track (set_tempo, 4/4) + one instrument track on channel 0."""
    tick = lambda sec: int(round(sec * bpm / 60.0 * RESOLUTION))  # noqa: E731
    ev = []
    for vel, pitch, start, end in notes:
        ev.append((tick(start), 1, pitch, vel))
        ev.append((tick(end), 0, pitch, 0))            # note-on with velocity 0 == note-off (pretty_midi's form)
    ev.sort(key=lambda e: (e[0], e[1], e[2]))           # at equal ticks: offs before ons
    last = max((e[0] for e in ev), default=0)
    tempo = int(round(6e7 / bpm))
    t0 = b"\x00\xff\x51\x03" + struct.pack(">I", tempo)[1:] + b"\x00\xff\x58\x04\x04\x02\x18\x08"
    t0 += _vlq(last + 1) + b"\xff\x2f\x00"
    t1 = b"\x00" + bytes([0xC0, program & 0x7F])
    cur = 0
    for tk, _, pitch, vel in ev:
        t1 += _vlq(tk - cur) + bytes([0x90, pitch & 0x7F, vel & 0x7F])
        cur = tk
    t1 += _vlq(1) + b"\xff\x2f\x00"
    with open(path, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, 1, 2, RESOLUTION))
        for trk in (t0, t1):
            f.write(b"MTrk" + struct.pack(">I", len(trk)) + trk)


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


def read_smf_notes(path: str):
    """Minimal parser of the files write_smf produces (tests): returns (division, tempo_us, [(tick_on, tick_off, pitch, vel)])."""
    data = open(path, "rb").read()
    assert data[:4] == b"MThd"
    _, fmt, ntrk, div = struct.unpack(">IHHH", data[4:14])
    pos, tempo, notes = 14, None, []
    for _ in range(ntrk):
        assert data[pos:pos + 4] == b"MTrk"
        ln = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        trk, p, tk, on = data[pos + 8:pos + 8 + ln], 0, 0, {}
        pos += 8 + ln
        while p < len(trk):
            d = 0
            while True:
                c = trk[p]; p += 1
                d = (d << 7) | (c & 0x7F)
                if not c & 0x80:
                    break
            tk += d
            st = trk[p]
            if st == 0xFF:
                mt, n = trk[p + 1], trk[p + 2]
                if mt == 0x51:
                    tempo = int.from_bytes(trk[p + 3:p + 6], "big")
                p += 3 + n
            elif st & 0xF0 == 0xC0:
                p += 2
            elif st & 0xF0 == 0x90:
                pitch, vel = trk[p + 1], trk[p + 2]
                p += 3
                if vel:
                    on.setdefault(pitch, []).append((tk, vel))
                else:
                    t_on, v = on[pitch].pop(0)
                    notes.append((t_on, tk, pitch, v))
            else:
                raise ValueError(f"unexpected status {st:#x}")
    return (fmt, div), tempo, sorted(notes)
