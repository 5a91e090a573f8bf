// make_rc_golden.js — golden wire frames for the rack controller's I2C register file (SURVEY.md §8f-4).
// The reference's Upconverter class (/root/reference/software/opi-rf-manager/lib/upconverter.js) cannot be loaded in the
// build container: its base class lib/i2c.js needs the npm packages i2c-bus and async-mutex, which are absent, and no
// stand-ins are written for them.  What the class does to BUILD a frame is three calls into Node's own Buffer
// (a dependency that IS here: node v12.22.9): Buffer.alloc(len + 1), writeUInt8(reg, 0), payload.copy(buf, 1)
// (upconverter.js:41-59), with payloads made by writeFloatLE / writeBigUInt64LE / a plain byte
// (:136-141, :146-156, :176-187, :233-244, :262-268).  This script makes those same Buffer calls on a list of values and
// records the bytes; it holds no reference code.  Run in the build container:  node tests/golden/make_rc_golden.js
"use strict";
const fs = require("fs");
const path = require("path");

function frame(reg, payload)            // the call sequence of upconverter.js:53-58
{
    const buf = Buffer.alloc(payload.length + 1, 0);
    buf.writeUInt8(reg, 0);
    payload.copy(buf, 1, 0);
    return buf;
}
function f32(v) { const b = Buffer.alloc(4); b.writeFloatLE(v, 0); return b; }
function u64(v) { const b = Buffer.alloc(8); b.writeBigUInt64LE(v, 0); return b; }

const out = [];
for(const [name, reg] of [["if_att", 0x20], ["rf1_att", 0x24], ["rf2_att", 0x28]])
    for(const v of [0, 0.25, 3.5, 10, 12.75, 31.75, 32.75, 1 / 3, 7.123456789])
        out.push({op: "write_f32", name, reg, value: v, frame: frame(reg, f32(v)).toString("hex")});
for(const v of [-10, -3.25, 0, 17.5, 40, 1e-3])
    out.push({op: "write_f32", name: "rf_out_pwr_low_thresh", reg: 0x34, value: v, frame: frame(0x34, f32(v)).toString("hex")});
for(const v of [35000000n, 1875000000n, 2400000000n, 4400000000n, 4294967296n, 739750000n])
    out.push({op: "write_u64", name: "lo_freq", reg: 0x10, value: v.toString(), frame: frame(0x10, u64(v)).toString("hex")});
for(const [reg, v] of [[0x01, 0x80], [0x01, 0x0F], [0x31, 0x13], [0x31, 0x00], [0xFF, 0xAA]])
    out.push({op: "write_u8", name: "byte", reg, value: v, frame: frame(reg, Buffer.from([v])).toString("hex")});
// read side: what the typed getters do with the bytes that come back (readFloatLE / readUInt32LE / readUInt16LE /
// readBigUInt64LE, upconverter.js:75-134, :188-257)
const rd = [];
for(const hex of ["0000803f", "cdcc4c3e", "00000000", "0000c07f", "ffffffff", "9a99f141", "d9ce17c1"])
{
    const b = Buffer.from(hex, "hex");
    const f = b.readFloatLE(0);
    rd.push({op: "read_f32", bytes: hex, value: Number.isNaN(f) ? "nan" : f, u32: b.readUInt32LE(0), u16: b.readUInt16LE(0)});
}
for(const hex of ["00f2052a01000000", "ffffffffffffffff", "c0e1e40000000000"])
    rd.push({op: "read_u64", bytes: hex, value: Buffer.from(hex, "hex").readBigUInt64LE(0).toString()});
{
    const b = Buffer.from("78563412f0debc9a", "hex");   // get_unique_id(): high word "-" low word, upper-case hex (:75-80)
    rd.push({op: "unique_id", bytes: b.toString("hex"),
             value: b.readUInt32LE(4).toString(16).toUpperCase() + "-" + b.readUInt32LE(0).toString(16).toUpperCase()});
}
fs.writeFileSync(path.join(__dirname, "rc_reg_golden.json"), JSON.stringify({node: process.version, writes: out, reads: rd}, null, 1));
console.log("wrote", out.length, "write frames and", rd.length, "read vectors");
