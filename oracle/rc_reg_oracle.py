"""TEST INFRASTRUCTURE — slave-side model of the upconverter's I2C register file.

Restates /root/reference/software/upconverter/src/main.c: the register map and its write/read masks (:33-55, :345-412),
the receive ISR (:446-513: first byte of a transfer = register pointer, every further byte a masked write with
post-increment of the one-byte pointer, change flags raised when the LAST byte of a multi-byte register arrives in a
transfer that carried at least the whole register) and the transmit ISR (:425-445: masked read with post-increment,
bit 0 of RF_OUT_PWR_STATUS cleared once it has been read).  Pure Python, used by tests/ only: the product's codec
(qo-100-tools_amd/csrc/rc_reg.c) is the HOST side of this wire and never sees this file.  Pinned by the frames
tests/golden/rc_reg_golden.json (made with Node's Buffer exactly as the daemon's call sites use it; the reference's
Upconverter class itself cannot be loaded here: its base class needs the absent npm packages i2c-bus and async-mutex)."""
import struct

REG = dict(STATUS=0x00, CONFIG=0x01, LO_FREQ=0x10, LO_REF_FREQ=0x18, LO_PFD_FREQ=0x1C, IF_ATT=0x20, RF1_ATT=0x24,
           RF2_ATT=0x28, RF_OUT_PWR_STATUS=0x30, RF_OUT_PWR_CONFIG=0x31, RF_OUT_PWR_LOW_THRESH=0x34, RF_OUT_PWR=0x38,
           VIN_VOLTAGE=0xC0, V5V0_VOLTAGE=0xC4, V5V0_CURRENT=0xC8, AVDD_VOLTAGE=0xD0, DVDD_VOLTAGE=0xD4,
           IOVDD_VOLTAGE=0xD8, CORE_VOLTAGE=0xDC, EMU_TEMP=0xE0, ADC_TEMP=0xE4, SW_VERSION=0xF4, DEV_UIDL=0xF8,
           DEV_UIDH=0xFC)
# (register, bytes, writable) as initialised at main.c:345-412; every listed register is fully readable
MASKS = [("STATUS", 1, False), ("CONFIG", 1, True), ("LO_FREQ", 8, True), ("LO_REF_FREQ", 4, False),
         ("LO_PFD_FREQ", 4, False), ("IF_ATT", 4, True), ("RF1_ATT", 4, True), ("RF2_ATT", 4, True),
         ("RF_OUT_PWR_CONFIG", 1, True), ("RF_OUT_PWR_LOW_THRESH", 4, True), ("RF_OUT_PWR", 4, False),
         ("VIN_VOLTAGE", 4, False), ("V5V0_VOLTAGE", 4, False), ("V5V0_CURRENT", 4, False), ("AVDD_VOLTAGE", 4, False),
         ("DVDD_VOLTAGE", 4, False), ("IOVDD_VOLTAGE", 4, False), ("CORE_VOLTAGE", 4, False), ("EMU_TEMP", 4, False),
         ("ADC_TEMP", 4, False), ("SW_VERSION", 2, False), ("DEV_UIDL", 4, False), ("DEV_UIDH", 4, False)]
# change flags of the receive ISR: (register, width) -> flag name (main.c:461-509)
FLAGS = {("LO_FREQ", 8): "lo_changed", ("IF_ATT", 4): "att_if", ("RF1_ATT", 4): "att_rf1", ("RF2_ATT", 4): "att_rf2",
         ("RF_OUT_PWR_CONFIG", 1): "power_meter", ("RF_OUT_PWR_LOW_THRESH", 4): "power_meter"}


class UpconverterSlave:
    def __init__(self):
        self.reg = bytearray(256)
        self.wmask = bytearray(256)
        self.rmask = bytearray(256)
        for name, width, writable in MASKS:
            a = REG[name]
            for i in range(width):
                self.rmask[a + i] = 0xFF
                self.wmask[a + i] = 0xFF if writable else 0x00
        # RF_OUT_PWR_STATUS has no mask lines in main.c: both masks stay at their zero initialisation
        self.pointer = 0
        self.count = 0
        self.flags = set()

    def start(self):                      # address match (main.c:415-424)
        self.count = 0

    def rx(self, byte):                   # main.c:446-513
        self.count += 1
        if self.count == 1:
            self.pointer = byte & 0xFF
            return
        p = self.pointer
        self.reg[p] = (self.reg[p] & ~self.wmask[p] & 0xFF) | (byte & self.wmask[p])
        self.pointer = (p + 1) & 0xFF
        for (name, width), flag in FLAGS.items():
            if self.pointer == (REG[name] + width) & 0xFF and (self.count - 1) >= width:
                self.flags.add(flag)

    def tx(self):                         # main.c:425-445
        self.count += 1
        p = self.pointer
        data = self.reg[p] & self.rmask[p]
        self.pointer = (p + 1) & 0xFF
        if self.pointer == REG["RF_OUT_PWR_STATUS"] + 1 and self.count >= 1:
            self.reg[REG["RF_OUT_PWR_STATUS"]] &= ~1 & 0xFF      # clear on read
        return data

    # whole transfers
    def write_transfer(self, frame):
        self.start()
        for b in frame:
            self.rx(b)

    def read_transfer(self, count):
        self.start()
        return bytes(self.tx() for _ in range(count))

    def get_f32(self, name):
        return struct.unpack_from("<f", self.reg, REG[name])[0]

    def get_u64(self, name):
        return struct.unpack_from("<Q", self.reg, REG[name])[0]
