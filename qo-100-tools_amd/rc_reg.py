"""ctypes mirror of include/rc_reg.h (host-side codec of the rack controllers' I2C register-file wire format)."""
import ctypes

from . import if_fir

EXPORTS = ["rc_reg_encode_write", "rc_reg_encode_read", "rc_reg_put_u16", "rc_reg_put_u32", "rc_reg_put_u64",
           "rc_reg_put_f32", "rc_reg_get_u16", "rc_reg_get_u32", "rc_reg_get_u64", "rc_reg_get_f32",
           "rc_upc_encode_attenuation", "rc_upc_encode_low_power_threshold", "rc_upc_encode_lo_frequency",
           "rc_upc_decode_unique_id", "rc_upc_if_attenuation_for_power"]
REG_IF_ATT, REG_RF1_ATT, REG_RF2_ATT, REG_LO_FREQ, REG_LOW_THRESH = 0x20, 0x24, 0x28, 0x10, 0x34
_u8p = ctypes.POINTER(ctypes.c_uint8)
_ready = False


def lib():
    global _ready
    L = if_fir.lib()
    if not _ready:
        L.rc_reg_encode_write.argtypes = [ctypes.c_uint8, _u8p, ctypes.c_uint32, _u8p, ctypes.c_uint32]
        L.rc_reg_encode_write.restype = ctypes.c_uint32
        L.rc_reg_encode_read.argtypes = [ctypes.c_uint8, _u8p, ctypes.c_uint32]
        L.rc_reg_encode_read.restype = ctypes.c_uint32
        L.rc_reg_get_f32.argtypes = [_u8p]
        L.rc_reg_get_f32.restype = ctypes.c_float
        L.rc_reg_get_u16.argtypes = [_u8p]
        L.rc_reg_get_u16.restype = ctypes.c_uint16
        L.rc_reg_get_u32.argtypes = [_u8p]
        L.rc_reg_get_u32.restype = ctypes.c_uint32
        L.rc_reg_get_u64.argtypes = [_u8p]
        L.rc_reg_get_u64.restype = ctypes.c_uint64
        L.rc_upc_encode_attenuation.argtypes = [ctypes.c_uint8, ctypes.c_float, _u8p]
        L.rc_upc_encode_attenuation.restype = ctypes.c_uint8
        L.rc_upc_encode_low_power_threshold.argtypes = [ctypes.c_float, _u8p]
        L.rc_upc_encode_low_power_threshold.restype = ctypes.c_uint8
        L.rc_upc_encode_lo_frequency.argtypes = [ctypes.c_uint64, _u8p]
        L.rc_upc_encode_lo_frequency.restype = ctypes.c_uint8
        L.rc_upc_decode_unique_id.argtypes = [_u8p, ctypes.c_char_p, ctypes.c_uint32]
        L.rc_upc_decode_unique_id.restype = ctypes.c_uint8
        L.rc_upc_if_attenuation_for_power.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_float,
                                                      ctypes.POINTER(ctypes.c_float)]
        L.rc_upc_if_attenuation_for_power.restype = ctypes.c_uint8
        _ready = True
    return L


def _buf(data):
    return (ctypes.c_uint8 * len(data))(*data)


def encode_write(reg, payload):
    """rc_reg_encode_write(): pointer byte + payload, or None when rejected."""
    out = (ctypes.c_uint8 * (len(payload) + 1))()
    n = lib().rc_reg_encode_write(int(reg), _buf(payload) if len(payload) else None, len(payload), out, len(out))
    return bytes(out[:n]) if n else None


def encode_read(reg):
    out = (ctypes.c_uint8 * 1)()
    return bytes(out[:lib().rc_reg_encode_read(int(reg), out, 1)])


def encode_attenuation(reg, att):
    out = (ctypes.c_uint8 * 5)()
    return bytes(out) if lib().rc_upc_encode_attenuation(int(reg), float(att), out) else None


def encode_low_power_threshold(power):
    out = (ctypes.c_uint8 * 5)()
    return bytes(out) if lib().rc_upc_encode_low_power_threshold(float(power), out) else None


def encode_lo_frequency(freq):
    out = (ctypes.c_uint8 * 9)()
    return bytes(out) if lib().rc_upc_encode_lo_frequency(int(freq), out) else None


def get_f32(data):
    return float(lib().rc_reg_get_f32(_buf(data)))


def get_u16(data):
    return int(lib().rc_reg_get_u16(_buf(data)))


def get_u32(data):
    return int(lib().rc_reg_get_u32(_buf(data)))


def get_u64(data):
    return int(lib().rc_reg_get_u64(_buf(data)))


def decode_unique_id(data):
    out = ctypes.create_string_buffer(32)
    return out.value.decode() if lib().rc_upc_decode_unique_id(_buf(data), out, 32) else None


def if_attenuation_for_power(mean_square, target_dbfs, current):
    nxt = ctypes.c_float(0)
    ok = lib().rc_upc_if_attenuation_for_power(float(mean_square), float(target_dbfs), float(current), ctypes.byref(nxt))
    return float(nxt.value) if ok else None
