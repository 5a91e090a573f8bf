"""CPU tests of the rack controller's register-file wire format: the host codec in libif_fir.so
(include/rc_reg.h) against frames made with Node's Buffer the way the daemon's call sites make them
(tests/golden/rc_reg_golden.json, tests/golden/make_rc_golden.js), and against the oracle's model of the slave
(oracle/rc_reg_oracle.py: pointer, masks, auto-increment, change flags, clear-on-read as in the firmware)."""
import json
import math
import os
import re
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "rc_reg_golden.json")))


@pytest.fixture(scope="module")
def rc(pkg):
    return pkg.rc_reg


@pytest.fixture()
def slave():
    from oracle import rc_reg_oracle
    return rc_reg_oracle.UpconverterSlave()


def test_library_exports_every_declared_symbol(rc):
    header = open(os.path.join(ROOT, "include", "rc_reg.h")).read()
    declared = set(re.findall(r"\b(rc_(?:reg|upc)_[a-z0-9_]+)\s*\(", header))
    assert declared == set(rc.EXPORTS), declared ^ set(rc.EXPORTS)
    for name in declared:
        assert hasattr(rc.lib(), name), name


def test_write_frames_equal_the_node_buffer_vectors(rc):
    assert len(GOLD["writes"]) >= 40
    for v in GOLD["writes"]:
        want = bytes.fromhex(v["frame"])
        if v["op"] == "write_f32":
            payload = struct.pack("<f", v["value"])
            typed = (rc.encode_low_power_threshold(v["value"]) if v["reg"] == rc.REG_LOW_THRESH
                     else rc.encode_attenuation(v["reg"], v["value"]))
            assert typed == want, v
        elif v["op"] == "write_u64":
            payload = struct.pack("<Q", int(v["value"]))
            assert rc.encode_lo_frequency(int(v["value"])) == want, v
        else:
            payload = bytes([v["value"]])
        assert rc.encode_write(v["reg"], payload) == want, v
        assert rc.encode_read(v["reg"]) == want[:1]


def test_read_payloads_decode_like_node(rc):
    for v in GOLD["reads"]:
        data = bytes.fromhex(v["bytes"])
        if v["op"] == "read_f32":
            got = rc.get_f32(data)
            assert (math.isnan(got) and v["value"] == "nan") or got == v["value"], v
            assert rc.get_u32(data) == v["u32"] and rc.get_u16(data) == v["u16"]
        elif v["op"] == "read_u64":
            assert rc.get_u64(data) == int(v["value"])
        else:
            assert rc.decode_unique_id(data) == v["value"]


def test_range_checks_mirror_the_daemon(rc):
    # upconverter.js throws "out of bounds" for these (:148-149, :178-179, :235-236): the codec returns 0
    for att in (-0.01, 32.76, float("nan"), 1e9):
        assert rc.encode_attenuation(rc.REG_IF_ATT, att) is None
    assert rc.encode_attenuation(0x30, 1.0) is None                   # not an attenuator register
    for p in (-10.01, 40.5, float("nan")):
        assert rc.encode_low_power_threshold(p) is None
    for f in (0, 34_999_999, 4_400_000_001, 1 << 63):
        assert rc.encode_lo_frequency(f) is None
    assert rc.encode_write(0xFE, b"\x01\x02\x03") is None             # would run past register 255
    assert rc.encode_write(0x20, b"") is None


def test_frames_drive_the_slave_model_like_the_firmware(rc, slave):
    slave.write_transfer(rc.encode_attenuation(rc.REG_IF_ATT, 12.75))
    assert slave.get_f32("IF_ATT") == 12.75 and slave.flags == {"att_if"}
    slave.flags.clear()
    slave.write_transfer(rc.encode_lo_frequency(1_875_000_000))
    assert slave.get_u64("LO_FREQ") == 1_875_000_000 and slave.flags == {"lo_changed"}
    # a short transfer that ends inside a register writes its bytes but raises no flag (main.c: count check)
    slave.flags.clear()
    slave.write_transfer(rc.encode_write(rc.REG_RF1_ATT, struct.pack("<f", 3.5)[:2]))
    assert not slave.flags and bytes(slave.reg[0x24:0x26]) == struct.pack("<f", 3.5)[:2]
    # auto-increment: one 12-byte transfer sets the three attenuators and raises the three flags
    slave.write_transfer(rc.encode_write(rc.REG_IF_ATT, struct.pack("<3f", 1.0, 2.25, 31.75)))
    assert (slave.get_f32("IF_ATT"), slave.get_f32("RF1_ATT"), slave.get_f32("RF2_ATT")) == (1.0, 2.25, 31.75)
    assert slave.flags == {"att_if", "att_rf1", "att_rf2"}
    # read-only registers ignore writes (write mask 0), unmapped ones read as 0 (read mask 0)
    slave.reg[0x38:0x3C] = struct.pack("<f", 21.5)
    slave.write_transfer(rc.encode_write(0x38, struct.pack("<f", -1.0)))
    slave.write_transfer(rc.encode_read(0x38))
    assert rc.get_f32(slave.read_transfer(4)) == 21.5
    slave.write_transfer(rc.encode_write(0x80, b"\xAA\xBB"))
    slave.write_transfer(rc.encode_read(0x80))
    assert slave.read_transfer(2) == b"\x00\x00"
    # a read runs on through the following registers; the pointer is one byte and wraps
    slave.write_transfer(rc.encode_read(rc.REG_IF_ATT))
    assert struct.unpack("<3f", slave.read_transfer(12)) == (1.0, 2.25, 31.75)
    slave.reg[0xFC:0x100] = b"\x11\x22\x33\x44"
    slave.reg[0x00] = 0x5A
    slave.write_transfer(rc.encode_read(0xFE))
    assert slave.read_transfer(3) == b"\x33\x44\x5A"
    # clear-on-read of the low-power latch (bit 0 of 0x30); the register has no read mask in the firmware: reads 0
    slave.reg[0x30] = 0x03
    slave.write_transfer(rc.encode_read(0x30))
    assert slave.read_transfer(1) == b"\x00" and slave.reg[0x30] == 0x02


def test_level_loop_step(rc):
    # -20 dBFS measured, -12 dBFS wanted: 8 dB less attenuation, in 0.25 dB steps, clamped to the attenuator's range
    assert rc.if_attenuation_for_power(0.01, -12.0, 20.0) == 12.0
    assert rc.if_attenuation_for_power(0.5, -12.0, 10.0) == pytest.approx(19.0, abs=1e-6)   # -3.01 dBFS -> +8.99 -> 19.0
    assert rc.if_attenuation_for_power(1e-9, -12.0, 5.0) == 0.0
    assert rc.if_attenuation_for_power(1.0, -40.0, 30.0) == 32.75
    assert rc.if_attenuation_for_power(0.0, -12.0, 5.0) is None and rc.if_attenuation_for_power(float("nan"), -12.0, 5.0) is None
    frame = rc.encode_attenuation(rc.REG_IF_ATT, rc.if_attenuation_for_power(0.01, -12.0, 20.0))
    assert frame == bytes([0x20]) + struct.pack("<f", 12.0)
