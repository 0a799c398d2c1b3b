/* msb_oracle.c -- CPU restatement of the fork's measurement wire format.  TEST INFRASTRUCTURE ONLY
 * (tests/ and nothing else may call it).
 *
 * Follows Source/TransformFunctions/msb_funcs.c: bit_writer :47-77, write_bb_msb :92-145,
 * read_bit :14-26, bit_reader :28-43, read_bb_msb :152-233 (its printf output left out), with the
 * structures of Include/msb_funcs.h:16-41.  The reference file itself is not compiled into
 * oracle/_ref: write_msb in the same file needs MSVC's fopen_s/errno_t, which this image lacks
 * (no stand-ins).  The reference holds no vector for this format: PARITY UNPINNED, the checks
 * are hand-derived bit patterns and the write -> read round trip. */
#include <stdint.h>
#include <string.h>

#define ORC_SYS_GPS 0x01
#define ORC_SYS_GAL 0x02

typedef struct {
  double time_tag;
  float pseudorange, code_phase, doppler, carr_phase, cno;
  uint16_t prn, constellation;
} orc_meas_sat;
typedef struct {
  uint64_t time;
  double sec;
  int num_sat;
  orc_meas_sat sats[20];
} orc_meas;

/* msb_funcs.c:47-77 */
static void orc_bit_writer(uint8_t *buff, int *bit_pos, uint64_t val, int num_bits)
{
  int bit_index = *bit_pos;
  for (int i = num_bits - 1; i >= 0; --i) {
    uint8_t bit = (uint8_t)((val >> i) & 0x1u);
    int byte_idx = bit_index >> 3;
    int bit_in_byte_from_msb = 7 - (bit_index & 7);
    if (bit) buff[byte_idx] |= (uint8_t)(1u << bit_in_byte_from_msb);
    else buff[byte_idx] &= (uint8_t)~(1u << bit_in_byte_from_msb);
    ++bit_index;
  }
  *bit_pos = bit_index;
}

/* msb_funcs.c:92-145; the caller provides a buffer that is large enough (the reference's is 2048) */
int orc_write_bb_msb(const orc_meas *m, uint8_t *bin_buff, int buff_size)
{
  int pos = 0;
  memset(bin_buff, 0, (size_t)buff_size);
  int64_t prn_mask = 0;
  for (int i = 0; i < m->num_sat; i++)
    if (m->sats[i].constellation == ORC_SYS_GPS) prn_mask |= (0x1LL << (m->sats[i].prn - 1));
  orc_bit_writer(bin_buff, &pos, (uint64_t)prn_mask, 32);
  for (int i = 0; i < m->num_sat; i++) {
    if (m->sats[i].constellation == ORC_SYS_GPS) {
      int64_t code_phase = (int64_t)(m->sats[i].code_phase / 9.54e-7f);
      int64_t doppler = (int64_t)(m->sats[i].doppler / 2.0f);
      int64_t cno = (int64_t)((m->sats[i].cno - 28) / 4);
      orc_bit_writer(bin_buff, &pos, (uint64_t)code_phase, 20);
      orc_bit_writer(bin_buff, &pos, (uint64_t)doppler, 13);
      orc_bit_writer(bin_buff, &pos, (uint64_t)cno, 4);
    }
  }
  int gal_present = 0;
  for (int i = 0; i < m->num_sat; i++)
    if (m->sats[i].constellation == ORC_SYS_GAL) { gal_present = 1; break; }
  orc_bit_writer(bin_buff, &pos, (uint64_t)gal_present, 3);
  if (gal_present) {
    int64_t gal_prn_mask = 0;
    for (int i = 0; i < m->num_sat; i++)
      if (m->sats[i].constellation == ORC_SYS_GAL) gal_prn_mask |= (0x1LL << (m->sats[i].prn - 1));
    orc_bit_writer(bin_buff, &pos, (uint64_t)gal_prn_mask, 50);
    for (int i = 0; i < m->num_sat; i++) {
      if (m->sats[i].constellation == ORC_SYS_GAL) {
        int64_t code_phase = (int64_t)(m->sats[i].code_phase / 9.54e-7f);
        int64_t doppler = (int64_t)(m->sats[i].doppler / 2.0f);
        int64_t cno = (int64_t)((m->sats[i].cno - 28) / 4);
        orc_bit_writer(bin_buff, &pos, (uint64_t)code_phase, 22);
        orc_bit_writer(bin_buff, &pos, (uint64_t)doppler, 13);
        orc_bit_writer(bin_buff, &pos, (uint64_t)cno, 4);
      }
    }
  }
  return pos / 8;
}

/* msb_funcs.c:14-43 with the reader state passed explicitly */
typedef struct { const uint8_t *buff; int cur_bit, buff_size; } orc_reader;
static int64_t orc_read_bit(orc_reader *r)
{
  if (r->cur_bit >= r->buff_size * 8) return -1;
  int64_t current_byte = r->buff[r->cur_bit / 8];
  int bit = (int)((current_byte >> (7 - r->cur_bit % 8)) & 0x01);
  r->cur_bit++;
  return bit;
}
static int64_t orc_bit_reader(orc_reader *r, unsigned num_bit, int is_signed)
{
  int64_t ans = 0;
  for (unsigned i = 0; i < num_bit; i++) ans = ans | (int64_t)((uint64_t)orc_read_bit(r) << (num_bit - (i + 1)));
  int64_t test = (0x1LL << (num_bit - 1)) & ans;
  if (test && is_signed) {
    int tmp = (int)(0xFFFFFFFFFFFFFFFFull << num_bit);  /* :39, an int on purpose */
    ans = ans | tmp;
  }
  return ans;
}

/* msb_funcs.c:152-233 */
void orc_read_bb_msb(const uint8_t *bin_buff, int bin_size, orc_meas *m)
{
  orc_reader r = {bin_buff, 0, bin_size};
  int64_t ans = orc_bit_reader(&r, 32, 0);
  int num_svs = 0, used_gps[32] = {0}, used_gal[50] = {0}, cnt = 0;
  for (int i = 0; i < 32; i++) {
    if (ans & 0x1) { num_svs++; used_gps[cnt++] = i + 1; }
    ans >>= 1;
  }
  m->num_sat = num_svs;
  for (int i = 0; i < num_svs; i++) {
    int64_t code_phase = orc_bit_reader(&r, 20, 0), doppler = orc_bit_reader(&r, 13, 1), cno = orc_bit_reader(&r, 4, 0);
    if (i >= 20) continue;
    m->sats[i].time_tag = (double)m->time + m->sec;
    m->sats[i].pseudorange = 0;
    m->sats[i].code_phase = ((float)code_phase) * 9.54e-7f;
    m->sats[i].cno = (float)(cno * 4 + 28);
    m->sats[i].doppler = (float)(2 * doppler);
    m->sats[i].prn = (uint16_t)used_gps[i];
    m->sats[i].constellation = ORC_SYS_GPS;
  }
  if (orc_bit_reader(&r, 3, 0) == 1) {
    ans = orc_bit_reader(&r, 50, 0);
    num_svs = 0; cnt = 0;
    for (int i = 0; i < 50; i++) {
      if (ans & 0x1) { num_svs++; used_gal[cnt++] = i + 1; }
      ans >>= 1;
    }
    int num_gps = m->num_sat;
    m->num_sat += num_svs;
    for (int i = 0; i < num_svs; i++) {
      int64_t code_phase = orc_bit_reader(&r, 22, 0), doppler = orc_bit_reader(&r, 13, 1), cno = orc_bit_reader(&r, 4, 0);
      if (num_gps + i >= 20) continue;
      m->sats[num_gps + i].time_tag = (double)m->time + m->sec;
      m->sats[num_gps + i].pseudorange = 0;
      m->sats[num_gps + i].code_phase = ((float)code_phase) * 9.54e-7f;
      m->sats[num_gps + i].cno = (float)(cno * 4 + 28);
      m->sats[num_gps + i].doppler = (float)(2 * doppler);
      m->sats[num_gps + i].prn = (uint16_t)used_gal[i];
      m->sats[num_gps + i].constellation = ORC_SYS_GAL;
    }
  }
}
