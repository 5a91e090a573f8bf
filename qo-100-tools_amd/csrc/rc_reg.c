/* rc_reg.c — host-side codec of the rack controllers' I2C register-file wire format (include/rc_reg.h).
 * Restates what /root/reference/software/opi-rf-manager/lib/upconverter.js:41-73 puts on the wire (pointer byte +
 * little-endian payload) and the daemon's range checks; the slave side (masks, auto-increment) lives in the oracle. */
#include "rc_reg.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#define RC_API __attribute__((visibility("default")))

RC_API uint32_t rc_reg_encode_write(uint8_t ubRegister, const uint8_t *pubPayload, uint32_t ulBytes, uint8_t *pubFrame,
                                    uint32_t ulFrameMax)
{
    if (!pubPayload || !pubFrame || !ulBytes || ulBytes + 1u > ulFrameMax || (uint32_t)ubRegister + ulBytes > 256u)
        return 0;
    pubFrame[0] = ubRegister;
    memcpy(pubFrame + 1, pubPayload, ulBytes);
    return ulBytes + 1u;
}

RC_API uint32_t rc_reg_encode_read(uint8_t ubRegister, uint8_t *pubFrame, uint32_t ulFrameMax)
{
    if (!pubFrame || !ulFrameMax)
        return 0;
    pubFrame[0] = ubRegister;
    return 1;
}

RC_API void rc_reg_put_u16(uint8_t *pub, uint16_t usValue)
{
    pub[0] = (uint8_t)usValue;
    pub[1] = (uint8_t)(usValue >> 8);
}
RC_API void rc_reg_put_u32(uint8_t *pub, uint32_t ulValue)
{
    for (int i = 0; i < 4; i++)
        pub[i] = (uint8_t)(ulValue >> (8 * i));
}
RC_API void rc_reg_put_u64(uint8_t *pub, uint64_t ullValue)
{
    for (int i = 0; i < 8; i++)
        pub[i] = (uint8_t)(ullValue >> (8 * i));
}
RC_API void rc_reg_put_f32(uint8_t *pub, float fValue)
{
    uint32_t u;
    memcpy(&u, &fValue, 4);
    rc_reg_put_u32(pub, u);
}
RC_API uint16_t rc_reg_get_u16(const uint8_t *pub) { return (uint16_t)(pub[0] | (pub[1] << 8)); }
RC_API uint32_t rc_reg_get_u32(const uint8_t *pub)
{
    return (uint32_t)pub[0] | ((uint32_t)pub[1] << 8) | ((uint32_t)pub[2] << 16) | ((uint32_t)pub[3] << 24);
}
RC_API uint64_t rc_reg_get_u64(const uint8_t *pub) { return (uint64_t)rc_reg_get_u32(pub) | ((uint64_t)rc_reg_get_u32(pub + 4) << 32); }
RC_API float rc_reg_get_f32(const uint8_t *pub)
{
    const uint32_t u = rc_reg_get_u32(pub);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static uint8_t encode_f32(uint8_t reg, float v, uint8_t *frame)
{
    uint8_t payload[4];
    rc_reg_put_f32(payload, v);
    return rc_reg_encode_write(reg, payload, 4, frame, 5) == 5;
}

RC_API uint8_t rc_upc_encode_attenuation(uint8_t ubRegister, float fAttenuation, uint8_t *pubFrame)
{
    if (ubRegister != RC_UPC_REG_IF_ATT && ubRegister != RC_UPC_REG_RF1_ATT && ubRegister != RC_UPC_REG_RF2_ATT)
        return 0;
    if (!pubFrame || isnan(fAttenuation) || fAttenuation < 0.0f || fAttenuation > 32.75f) /* upconverter.js:178-179 */
        return 0;
    return encode_f32(ubRegister, fAttenuation, pubFrame);
}

RC_API uint8_t rc_upc_encode_low_power_threshold(float fPower, uint8_t *pubFrame)
{
    if (!pubFrame || isnan(fPower) || fPower < -10.0f || fPower > 40.0f) /* upconverter.js:148-149 */
        return 0;
    return encode_f32(RC_UPC_REG_RF_OUT_PWR_LOW_THRESH, fPower, pubFrame);
}

RC_API uint8_t rc_upc_encode_lo_frequency(uint64_t ullFrequency, uint8_t *pubFrame)
{
    if (!pubFrame || ullFrequency < 35000000ull || ullFrequency > 4400000000ull) /* upconverter.js:235-236 */
        return 0;
    uint8_t payload[8];
    rc_reg_put_u64(payload, ullFrequency);
    return rc_reg_encode_write(RC_UPC_REG_LO_FREQ, payload, 8, pubFrame, 9) == 9;
}

RC_API uint8_t rc_upc_decode_unique_id(const uint8_t *pubData, char *pszOut, uint32_t ulOutBytes)
{
    if (!pubData || !pszOut || ulOutBytes < 18)
        return 0;
    /* toString(16).toUpperCase(): no leading zeros */
    snprintf(pszOut, ulOutBytes, "%X-%X", (unsigned)rc_reg_get_u32(pubData + 4), (unsigned)rc_reg_get_u32(pubData));
    return 1;
}

RC_API uint8_t rc_upc_if_attenuation_for_power(double dMeanSquare, double dTargetDbfs, float fCurrent, float *pfNext)
{
    if (!pfNext || !(dMeanSquare > 0.0) || !isfinite(dMeanSquare) || !isfinite(dTargetDbfs) || isnan(fCurrent))
        return 0;
    double next = (double)fCurrent + (10.0 * log10(dMeanSquare) - dTargetDbfs);
    next = floor(next * 4.0 + 0.5) / 4.0; /* 0.25 dB steps */
    if (next < 0.0)
        next = 0.0;
    if (next > 32.75)
        next = 32.75;
    *pfNext = (float)next;
    return 1;
}
