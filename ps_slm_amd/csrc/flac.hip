// Host-side FLAC decoder (no device code) for the dataset path: the reference loads ``.flac`` entries with torchaudio
// (Multitask/dataset/speech_dataset_large.py:123-127: waveform [C, T] float in [-1, 1), channels averaged); torchaudio and libFLAC
// are not on the image, so the published FLAC format is decoded here: STREAMINFO, frame headers (CRC-8), CONSTANT / VERBATIM /
// FIXED / LPC subframes with wasted bits, Rice / Rice2 residuals with escape partitions, independent / left-side / right-side /
// mid-side stereo, frame CRC-16, and the STREAMINFO MD5 of the decoded PCM (so a real file verifies its own decode).
// PARITY UNPINNED against libFLAC / torchaudio (absent): tests round-trip streams written by tests/flac_fixtures.py.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/tasu_hip.h"

#define TASU_OK 0
#define TASU_ERR_ARG 1

namespace {

constexpr int ERR_STREAM = 3;

struct BitReader {
  const uint8_t* p;
  int64_t n, pos = 0;      // pos in bits
  bool fail = false;
  uint32_t bits(int k) {   // k <= 32
    uint32_t v = 0;
    for (int i = 0; i < k; ++i) {
      const int64_t byte = pos >> 3;
      if (byte >= n) {
        fail = true;
        return 0;
      }
      v = (v << 1) | ((p[byte] >> (7 - (pos & 7))) & 1u);
      ++pos;
    }
    return v;
  }
  int32_t sbits(int k) {
    if (k == 0) return 0;
    const uint32_t v = bits(k);
    return k == 32 ? (int32_t)v : (int32_t)(v << (32 - k)) >> (32 - k);
  }
  uint32_t unary() {       // number of 0 bits before the next 1
    uint32_t q = 0;
    while (!fail && bits(1) == 0) ++q;
    return q;
  }
  void align() { pos = (pos + 7) & ~7ll; }
};

uint8_t crc8(const uint8_t* d, int64_t n) {
  uint8_t c = 0;
  for (int64_t i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
  }
  return c;
}
uint16_t crc16(const uint8_t* d, int64_t n) {
  uint16_t c = 0;
  for (int64_t i = 0; i < n; ++i) {
    c ^= (uint16_t)d[i] << 8;
    for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x8005) : (uint16_t)(c << 1);
  }
  return c;
}

// ---- MD5 (RFC 1321) of the decoded PCM, compared with STREAMINFO
struct Md5 {
  uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
  uint8_t buf[64];
  uint64_t len = 0;
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; ++i) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) f = (B & C) | (~B & D), g = i;
      else if (i < 32) f = (D & B) | (~D & C), g = (5 * i + 1) & 15;
      else if (i < 48) f = B ^ C ^ D, g = (3 * i + 5) & 15;
      else f = C ^ (B | ~D), g = (7 * i) & 15;
      const uint32_t t = D;
      D = C;
      C = B;
      B = B + rol(A + f + K[i] + m[g], S[i]);
      A = t;
    }
    a += A, b += B, c += C, d += D;
  }
  void update(const uint8_t* p, size_t n) {
    size_t fill = len & 63;
    len += n;
    while (n) {
      const size_t take = n < 64 - fill ? n : 64 - fill;
      memcpy(buf + fill, p, take);
      fill += take, p += take, n -= take;
      if (fill == 64) {
        block(buf);
        fill = 0;
      }
    }
  }
  void final(uint8_t out[16]) {
    const uint64_t bits = len * 8;
    const uint8_t pad = 0x80, zero = 0;
    update(&pad, 1);
    while ((len & 63) != 56) update(&zero, 1);
    uint8_t l[8];
    for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
    update(l, 8);
    const uint32_t v[4] = {a, b, c, d};
    for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
  }
};

struct Info {
  int rate = 0, channels = 0, bps = 0;
  int64_t total = 0;
  uint8_t md5[16] = {0};
  int64_t first_frame = 0;   // byte offset of the first audio frame
};

int parse_header(const uint8_t* d, int64_t n, Info& in) {
  if (n < 42 || memcmp(d, "fLaC", 4) != 0) return ERR_STREAM;
  int64_t off = 4;
  bool have = false;
  for (;;) {
    if (off + 4 > n) return ERR_STREAM;
    const bool last = d[off] & 0x80;
    const int type = d[off] & 0x7f;
    const int64_t len = ((int64_t)d[off + 1] << 16) | (d[off + 2] << 8) | d[off + 3];
    off += 4;
    if (off + len > n) return ERR_STREAM;
    if (type == 0) {
      if (len < 34) return ERR_STREAM;
      const uint8_t* s = d + off;
      in.rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      in.channels = ((s[12] >> 1) & 7) + 1;
      in.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      in.total = ((int64_t)(s[13] & 15) << 32) | ((int64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
      memcpy(in.md5, s + 18, 16);
      have = true;
    }
    off += len;
    if (last) break;
  }
  in.first_frame = off;
  return have && in.rate > 0 && in.bps >= 4 && in.bps <= 32 ? 0 : ERR_STREAM;
}

bool read_residual(BitReader& br, int32_t* res, int blocksize, int order) {
  const int method = (int)br.bits(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  const int porder = (int)br.bits(4);
  const int parts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
  int idx = 0;
  for (int p = 0; p < parts && !br.fail; ++p) {
    int cnt = (blocksize >> porder) - (p == 0 ? order : 0);
    if (porder == 0) cnt = blocksize - order;
    if (cnt < 0) return false;
    const int k = (int)br.bits(pbits);
    if (k == esc) {
      const int nb = (int)br.bits(5);
      for (int i = 0; i < cnt; ++i) res[idx++] = br.sbits(nb);
    } else {
      for (int i = 0; i < cnt && !br.fail; ++i) {
        const uint32_t q = br.unary();
        const uint32_t u = (q << k) | (k ? br.bits(k) : 0);
        res[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
  }
  return !br.fail && idx == blocksize - order;
}

bool read_subframe(BitReader& br, int64_t* out, int blocksize, int bps) {
  if (br.bits(1) != 0) return false;
  const int type = (int)br.bits(6);
  int wasted = 0;
  if (br.bits(1)) wasted = (int)br.unary() + 1;
  bps -= wasted;
  if (bps <= 0) return false;
  std::vector<int32_t> res;
  if (type == 0) {
    const int64_t v = br.sbits(bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) out[i] = br.sbits(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = br.sbits(bps);
    res.resize(blocksize);
    if (!read_residual(br, res.data(), blocksize, order)) return false;
    for (int i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      switch (order) {
        case 1: pred = out[i - 1]; break;
        case 2: pred = 2 * out[i - 1] - out[i - 2]; break;
        case 3: pred = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
        case 4: pred = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
        default: break;
      }
      out[i] = pred + res[i - order];
    }
  } else if (type >= 32) {
    const int order = type - 31;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = br.sbits(bps);
    const int prec = (int)br.bits(4) + 1;
    if (prec == 16) return false;
    const int shift = br.sbits(5);
    if (shift < 0) return false;
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
    res.resize(blocksize);
    if (!read_residual(br, res.data(), blocksize, order)) return false;
    for (int i = order; i < blocksize; ++i) {
      int64_t acc = 0;
      for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * out[i - 1 - j];
      out[i] = (acc >> shift) + res[i - order];
    }
  } else {
    return false;
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) out[i] <<= wasted;
  return !br.fail;
}

// decodes every frame; sink(channel-interleaved samples of one block)
template <class Sink>
int decode_stream(const uint8_t* d, int64_t n, const Info& in, Sink&& sink) {
  BitReader br{d, n};
  br.pos = in.first_frame * 8;
  std::vector<int64_t> ch[8];
  int64_t done = 0;
  while ((br.pos >> 3) + 2 <= n && (in.total == 0 || done < in.total)) {
    const int64_t start = br.pos >> 3;
    if (br.bits(14) != 0x3ffe) {
      // STREAMINFO without a total: the stream ends where the frames end (trailing bytes such as an ID3v1 tag are not frames)
      if (in.total == 0 && done > 0) break;
      return ERR_STREAM;
    }
    br.bits(1);
    br.bits(1);                                    // blocking strategy: the coded number is not needed for sequential decode
    const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ca = (int)br.bits(4), ss_code = (int)br.bits(3);
    if (br.bits(1) != 0) return ERR_STREAM;
    // UTF-8-like coded frame / sample number
    uint32_t first = br.bits(8);
    int extra = 0;
    if (first & 0x80) {
      while (first & (0x80u >> (extra + 1))) ++extra;      // leading ones after the first = continuation bytes
      if (extra < 1 || extra > 6) return ERR_STREAM;
      for (int i = 0; i < extra; ++i) br.bits(8);
    }
    int blocksize;
    if (bs_code == 0) return ERR_STREAM;
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.bits(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.bits(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.bits(8);
    else if (sr_code == 13 || sr_code == 14) br.bits(16);
    else if (sr_code == 15) return ERR_STREAM;
    static const int ss_tab[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = ss_tab[ss_code] == 0 ? in.bps : ss_tab[ss_code];
    if (bps < 0) return ERR_STREAM;
    const int64_t hdr_end = br.pos >> 3;
    const uint8_t want8 = (uint8_t)br.bits(8);
    if (br.fail || crc8(d + start, hdr_end - start) != want8) return ERR_STREAM;
    if (ca > 10) return ERR_STREAM;                 // channel assignments 11-15 are reserved
    const int nch = ca < 8 ? ca + 1 : 2;
    if (nch != in.channels) return ERR_STREAM;
    if (ca >= 8 && bps + 1 > 32) return ERR_STREAM; // a 33-bit side channel is beyond the 32-bit reads of BitReader
    for (int c = 0; c < nch; ++c) {
      ch[c].assign(blocksize, 0);
      const bool side = (ca == 8 && c == 1) || (ca == 9 && c == 0) || (ca == 10 && c == 1);
      if (!read_subframe(br, ch[c].data(), blocksize, bps + (side ? 1 : 0))) return ERR_STREAM;
    }
    br.align();
    const int64_t body_end = br.pos >> 3;
    const uint16_t want16 = (uint16_t)br.bits(16);
    if (br.fail || crc16(d + start, body_end - start) != want16) return ERR_STREAM;
    if (ca == 8) {                                  // left, side = left - right
      for (int i = 0; i < blocksize; ++i) ch[1][i] = ch[0][i] - ch[1][i];
    } else if (ca == 9) {                           // side, right
      for (int i = 0; i < blocksize; ++i) ch[0][i] = ch[0][i] + ch[1][i];
    } else if (ca == 10) {                          // mid, side
      for (int i = 0; i < blocksize; ++i) {
        const int64_t s = ch[1][i];
        const int64_t m = (ch[0][i] << 1) | (s & 1);
        ch[0][i] = (m + s) >> 1;
        ch[1][i] = (m - s) >> 1;
      }
    }
    int take = blocksize;
    if (in.total && done + take > in.total) take = (int)(in.total - done);
    sink(ch, nch, take);
    done += take;
  }
  return in.total == 0 || done == in.total ? 0 : ERR_STREAM;
}

}  // namespace

extern "C" int tasu_flac_info(const uint8_t* data, int64_t n, int32_t* rate_channels_bps, int64_t* total_samples) {
  if (!data || !rate_channels_bps || !total_samples || n <= 0) return TASU_ERR_ARG;
  Info in;
  const int rc = parse_header(data, n, in);
  if (rc) return rc;
  rate_channels_bps[0] = in.rate, rate_channels_bps[1] = in.channels, rate_channels_bps[2] = in.bps;
  *total_samples = in.total;
  return TASU_OK;
}

extern "C" int tasu_flac_decode(const uint8_t* data, int64_t n, float* mono_out, int64_t capacity, int64_t* n_decoded) {
  if (!data || !mono_out || !n_decoded || n <= 0 || capacity < 0) return TASU_ERR_ARG;
  Info in;
  int rc = parse_header(data, n, in);
  if (rc) return rc;
  Md5 md5;
  const int bytes = (in.bps + 7) / 8;
  const float norm = 1.0f / (float)(1ll << (in.bps - 1));
  int64_t written = 0;
  bool overflow = false;
  std::vector<uint8_t> pcm;
  rc = decode_stream(data, n, in, [&](std::vector<int64_t>* ch, int nch, int take) {
    pcm.resize((size_t)take * nch * bytes);
    size_t o = 0;
    for (int i = 0; i < take; ++i)
      for (int c = 0; c < nch; ++c) {
        const int64_t v = ch[c][i];
        for (int b = 0; b < bytes; ++b) pcm[o++] = (uint8_t)(v >> (8 * b));
      }
    md5.update(pcm.data(), pcm.size());
    for (int i = 0; i < take; ++i) {
      if (written >= capacity) {
        overflow = true;
        return;
      }
      float s = 0.f;                               // torchaudio: every channel scaled to [-1, 1), then the channel mean
      for (int c = 0; c < nch; ++c) s += (float)ch[c][i] * norm;
      mono_out[written++] = nch > 1 ? s / (float)nch : s;
    }
  });
  *n_decoded = written;
  if (rc) return rc;
  if (overflow) return TASU_ERR_ARG;
  static const uint8_t zero[16] = {0};
  if (memcmp(in.md5, zero, 16) != 0) {
    uint8_t got[16];
    md5.final(got);
    if (memcmp(got, in.md5, 16) != 0) return ERR_STREAM;
  }
  return TASU_OK;
}
