/* rc_reg.h — wire format of the rack controllers' I2C register files: host-side codec.
 *
 * Every controller of the reference rack (upconverter, PA bias, LNB, relays, ...) is an I2C slave with a 256-byte
 * register file: a write transfer is ONE register-pointer byte followed by the payload, little-endian, the pointer
 * auto-increments per byte (/root/reference/software/upconverter/src/main.c:446-513: first byte = pointer, then masked
 * byte writes); a read is a pointer write followed by a read transfer (main.c:425-445).  The daemon builds the frames in
 * /root/reference/software/opi-rf-manager/lib/upconverter.js:41-73 (Buffer.alloc(len + 1), writeUInt8(reg, 0),
 * payload.copy(buf, 1)) with payloads from writeFloatLE / writeBigUInt64LE (:176-187, :233-244).
 *
 * This header is what a C host that steers the rack from the filter's measurements (if_fir_power_device -> IF attenuator
 * register 0x20) would use instead of the Node classes.  Pure host code: no I2C access, no GPU; the caller owns the bus.
 * Conventions as in if_fir.h (uint8_t 1 = ok, 0 = rejected; Hungarian prefixes). */
#ifndef RC_REG_H
#define RC_REG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* upconverter register map (main.c:33-55); payload types in the comments there */
#define RC_UPC_REG_STATUS 0x00u
#define RC_UPC_REG_CONFIG 0x01u
#define RC_UPC_REG_LO_FREQ 0x10u               /* uint64, Hz */
#define RC_UPC_REG_IF_ATT 0x20u                /* float32, dB */
#define RC_UPC_REG_RF1_ATT 0x24u
#define RC_UPC_REG_RF2_ATT 0x28u
#define RC_UPC_REG_RF_OUT_PWR_STATUS 0x30u
#define RC_UPC_REG_RF_OUT_PWR_CONFIG 0x31u
#define RC_UPC_REG_RF_OUT_PWR_LOW_THRESH 0x34u /* float32, dBm */
#define RC_UPC_REG_RF_OUT_PWR 0x38u            /* float32, dBm, read-only */

/* write transfer = pointer byte + payload: returns the frame length (ulBytes + 1), 0 if it does not fit or ulBytes == 0
 * or the payload would run past register 255 (the pointer is one byte) */
uint32_t rc_reg_encode_write(uint8_t ubRegister, const uint8_t *pubPayload, uint32_t ulBytes, uint8_t *pubFrame,
                             uint32_t ulFrameMax);
/* pointer write that precedes a read transfer: 1 byte */
uint32_t rc_reg_encode_read(uint8_t ubRegister, uint8_t *pubFrame, uint32_t ulFrameMax);
/* little-endian payload fields */
void rc_reg_put_u16(uint8_t *pub, uint16_t usValue);
void rc_reg_put_u32(uint8_t *pub, uint32_t ulValue);
void rc_reg_put_u64(uint8_t *pub, uint64_t ullValue);
void rc_reg_put_f32(uint8_t *pub, float fValue);
uint16_t rc_reg_get_u16(const uint8_t *pub);
uint32_t rc_reg_get_u32(const uint8_t *pub);
uint64_t rc_reg_get_u64(const uint8_t *pub);
float rc_reg_get_f32(const uint8_t *pub);
/* typed upconverter frames with the daemon's range checks (upconverter.js:176-187, :146-156, :233-244):
 * attenuation 0 .. 32.75 dB into ubRegister = IF/RF1/RF2_ATT; threshold -10 .. 40 dBm; LO 35 MHz .. 4.4 GHz.
 * pubFrame: 5, 5 and 9 bytes.  0 = out of bounds / NaN / wrong register. */
uint8_t rc_upc_encode_attenuation(uint8_t ubRegister, float fAttenuation, uint8_t *pubFrame);
uint8_t rc_upc_encode_low_power_threshold(float fPower, uint8_t *pubFrame);
uint8_t rc_upc_encode_lo_frequency(uint64_t ullFrequency, uint8_t *pubFrame);
/* "HIGH-LOW" upper-case hex of the 8 bytes at register 0xF8 (upconverter.js:75-80); pszOut >= 18 bytes */
uint8_t rc_upc_decode_unique_id(const uint8_t *pubData, char *pszOut, uint32_t ulOutBytes);
/* BUILD-DEFINED level loop step (no reference counterpart): the IF attenuation that brings the filter's in-band power
 * (dMeanSquare = if_fir_power_device, full scale 1.0) to dTargetDbfs, from the current setting, in the attenuator's 0.25 dB
 * steps, clamped to 0 .. 32.75 dB.  Returns 0 for a non-positive or non-finite power. */
uint8_t rc_upc_if_attenuation_for_power(double dMeanSquare, double dTargetDbfs, float fCurrent, float *pfNext);

#ifdef __cplusplus
}
#endif
#endif
