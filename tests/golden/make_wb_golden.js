// make_wb_golden.js — captures golden vectors for the WB-transponder signal detector (SURVEY.md §8f-3) by RUNNING the
// reference's own routine: WBSpectrumMonitor.detect_signals() of
// /root/reference/software/opi-rf-manager/lib/wb_spectrum_monitor.js (static method, no socket is opened).
// Run in the build container only (node v12, the reference tree is not on the GPU box):
//     node tests/golden/make_wb_golden.js
// Output: tests/golden/wb_detect_golden.json = {frames: [{bins: base64 of uint16LE magnitudes, result | error}]}.
// This script holds the synthetic frame generator and nothing of the reference's code.
"use strict";
const path = require("path");
const fs = require("fs");
const Monitor = require("/root/reference/software/opi-rf-manager/lib/wb_spectrum_monitor.js");

let state = 0x51303031;
function rnd() // xorshift32, uniform in [0, 1)
{
    state ^= state << 13; state >>>= 0;
    state ^= state >>> 17;
    state ^= state << 5; state >>>= 0;
    return state / 4294967296;
}
function frame(len, noiseLo, noiseHi, signals)
{
    const v = new Array(len);
    for(let i = 0; i < len; i++)
        v[i] = noiseLo + rnd() * (noiseHi - noiseLo);
    for(const s of signals)
    {
        for(let i = Math.max(0, s.start); i < Math.min(len, s.start + s.width); i++)
        {
            const edge = Math.min(i - s.start, s.start + s.width - 1 - i);       // distance to the nearer edge
            const ramp = s.ramp > 0 ? Math.min(1, (edge + 1) / s.ramp) : 1;       // trapezoid
            v[i] = Math.max(v[i], s.floor + (s.power - s.floor) * ramp + (rnd() - 0.5) * s.ripple);
        }
    }
    const b = Buffer.alloc(len * 2);
    for(let i = 0; i < len; i++)
        b.writeUInt16LE(Math.max(0, Math.min(65535, Math.round(v[i]))), i * 2);
    return b;
}
const frames = [];
const beacon = {start: 60, width: 150, power: 42000, floor: 15000, ramp: 6, ripple: 900};
// hand-made cases
frames.push(frame(918, 8000, 12000, []));                                                           // noise only
frames.push(frame(918, 8000, 12000, [beacon]));
frames.push(frame(918, 8000, 12000, [beacon, {start: 300, width: 34, power: 30000, floor: 14000, ramp: 3, ripple: 500}]));
frames.push(frame(918, 8000, 12000, [beacon, {start: 700, width: 218, power: 41900, floor: 16000, ramp: 4, ripple: 300}])); // to the end, over-powered
frames.push(frame(918, 8000, 12000, [{start: 2, width: 40, power: 26000, floor: 12000, ramp: 0, ripple: 0}]));              // starts at the first averaged bin
frames.push(frame(918, 8000, 12000, [{start: 0, width: 918, power: 30000, floor: 30000, ramp: 0, ripple: 2000}]));          // all signal
frames.push(frame(918, 16400, 16600, []));                                                          // noise riding on the threshold
frames.push(frame(918, 8000, 12000, [beacon, {start: 400, width: 2, power: 50000, floor: 50000, ramp: 0, ripple: 0},
                                     {start: 500, width: 5, power: 20000, floor: 20000, ramp: 0, ripple: 0}]));          // too narrow: dropped
frames.push(frame(918, 0, 1, []));
frames.push(frame(918, 65535, 65535, []));
for(const len of [1, 2, 3, 4, 7, 100, 459, 1024, 4096])
    frames.push(frame(len, 8000, 12000, [{start: Math.floor(len / 3), width: Math.floor(len / 4), power: 33000, floor: 15000, ramp: 2, ripple: 400}]));
// random mixes
for(let k = 0; k < 44; k++)
{
    const sigs = rnd() < 0.8 ? [Object.assign({}, beacon, {power: 30000 + rnd() * 25000, start: 40 + Math.floor(rnd() * 60)})] : [];
    const count = Math.floor(rnd() * 7);
    for(let c = 0; c < count; c++)
        sigs.push({start: 230 + Math.floor(rnd() * 680), width: 2 + Math.floor(Math.pow(rnd(), 2) * 260),
                   power: 16000 + rnd() * 45000, floor: 10000 + rnd() * 8000, ramp: Math.floor(rnd() * 12), ripple: rnd() * 3000});
    frames.push(frame(918, 6000 + rnd() * 4000, 10000 + rnd() * 5000, sigs));
}
const out = {source: "WBSpectrumMonitor.detect_signals, /root/reference/software/opi-rf-manager/lib/wb_spectrum_monitor.js:36-180", frames: []};
let thrown = 0;
for(const b of frames)
{
    const rec = {bins: b.toString("base64")};
    try
    {
        rec.result = Monitor.detect_signals(b);
    }
    catch(e)
    {
        rec.error = String(e.code || e.name);
        thrown++;
    }
    out.frames.push(rec);
}
// non-finite numbers would become null in JSON: make them explicit
const text = JSON.stringify(out, (k, v) => (typeof v === "number" && !isFinite(v)) ? String(v) : v);
fs.writeFileSync(path.join(__dirname, "wb_detect_golden.json"), text);
console.log("frames", frames.length, "thrown", thrown, "bytes", text.length);
