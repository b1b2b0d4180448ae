// Host-side audio front end of `load_audio_from_file` (jukebox/main.py:29-45).  No device code: this file is the part of
// `lr.load(fpath, sr=44100)` (jukebox/main.py:31) that is arithmetic -- the sample-rate conversion -- as a plain C entry
// point on HOST pointers, so that the file-level boundary does not depend on a Python resampling package.
//
// What the reference runs there.  docker/jukebox-embed.dockerfile:42-58 builds the image of jukebox/main.py from
// python 3.7 + `pip install -e jukebox` (openai/jukebox @ 08efbbc, whose requirements pin librosa==0.7.2 and
// numba==0.48.0); librosa 0.7.2's `load` resamples with res_type="kaiser_best" = resampy's band-limited sinc
// interpolation (J. O. Smith, "Digital Audio Resampling"; resampy/core.py + interpn.py, dataflow-requirements.txt:4
// lists the package).  resampy is not under /root/reference and not installed here: the loop below restates its published
// algorithm (PARITY UNPINNED, like the rest of the Jukebox half):
//   for every output sample t:   n = int(time), frac = scale * (time - n)
//     left wing :  y[t] += (win[off + i*step] + eta * dwin[off + i*step]) * x[n - i]        i = 0 .. while inside win and x
//     right wing:  the same with frac -> scale - frac over x[n + 1 + k]
//     time += 1 / ratio                          (accumulated, as the original does -- not t / ratio)
// with scale = min(1, ratio), step = int(scale * num_table), off = int(frac * num_table), eta the remainder.  Arithmetic as
// in the original: filter weight in double, the running sum rounded to float after every tap (numba's `y[t] += w * x` on a
// float32 array).
#include <cstdint>
#include <vector>

#include "common.h"

using namespace llark;

// x [n_in] float (host) -> y [n_out] float (host, overwritten).  win / dwin [nwin] double (host): right half of the
// interpolation window sampled `num_table` times per zero crossing, already multiplied by `ratio` when ratio < 1, and its
// forward difference (last entry 0).  ratio = sr_new / sr_orig.
extern "C" int llark_resample_sinc_host(const float* x, int64_t n_in, double ratio, const double* win, const double* dwin, int nwin, int num_table,
                                        float* y, int64_t n_out) {
    LLARK_REQUIRE(x && win && dwin && y, "resample_sinc_host: null pointer");
    LLARK_REQUIRE(n_in > 0 && n_out > 0 && nwin > 1 && num_table > 0, "resample_sinc_host: empty input, output or filter");
    LLARK_REQUIRE(ratio > 0.0, "resample_sinc_host: ratio must be positive, got %g", ratio);
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const double time_increment = 1.0 / ratio;
    const int index_step = (int)(scale * num_table);
    LLARK_REQUIRE(index_step > 0, "resample_sinc_host: ratio %g too small for a table of %d samples per zero crossing", ratio, num_table);
    LLARK_REQUIRE((double)(n_out - 1) * time_increment < (double)n_in, "resample_sinc_host: %lld outputs at ratio %g overrun %lld inputs",
                  (long long)n_out, ratio, (long long)n_in);
    double time_register = 0.0;
    for (int64_t t = 0; t < n_out; ++t) {
        const int64_t n = (int64_t)time_register;
        double frac = scale * (time_register - (double)n);
        double index_frac = frac * num_table;
        int offset = (int)index_frac;
        double eta = index_frac - offset;
        float acc = 0.0f;
        int64_t i_max = (nwin - offset) / index_step;
        if (n + 1 < i_max) i_max = n + 1;
        for (int64_t i = 0; i < i_max; ++i) {
            const double w = win[offset + i * index_step] + eta * dwin[offset + i * index_step];
            acc = (float)((double)acc + w * (double)x[n - i]);
        }
        frac = scale - frac;
        index_frac = frac * num_table;
        offset = (int)index_frac;
        eta = index_frac - offset;
        int64_t k_max = (nwin - offset) / index_step;
        if (n_in - n - 1 < k_max) k_max = n_in - n - 1;
        for (int64_t k = 0; k < k_max; ++k) {
            const double w = win[offset + k * index_step] + eta * dwin[offset + k * index_step];
            acc = (float)((double)acc + w * (double)x[n + k + 1]);
        }
        y[t] = acc;
        time_register += time_increment;
    }
    return LLARK_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// FLAC.  `lr.load` (jukebox/main.py:31) and `sf.read` (m2t/gcs_utils.py:123) both go through libsndfile, which reads FLAC next to wav
// with no further dependency -- the one other container the reference's two file entries accept as they are.  A native decoder of
// the format as published (xiph.org FLAC format / RFC 9639): STREAMINFO, frame headers (CRC-8), CONSTANT / VERBATIM / FIXED / LPC
// subframes with wasted bits, partitioned Rice residuals with escape partitions, the three stereo decorrelations, frame CRC-16, and
// the MD5 signature of the decoded samples that STREAMINFO carries: a FLAC stream certifies its own decode, so "bit-equal to
// libFLAC / libsndfile" is checked at run time on every file rather than assumed.  Host code, host pointers.
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

struct BitReader {
    const uint8_t* d;
    int64_t n, pos;          // pos in BITS
    bool fail;
    uint32_t bits(int k) {   // k <= 32
        if (k == 0) return 0;
        if (pos + k > n * 8) {
            fail = true;
            pos = n * 8;
            return 0;
        }
        uint64_t v = 0;
        int got = 0;
        while (got < k) {
            const int64_t byte = pos >> 3;
            const int off = (int)(pos & 7), take = (8 - off) < (k - got) ? (8 - off) : (k - got);
            v = (v << take) | ((d[byte] >> (8 - off - take)) & ((1u << take) - 1));
            pos += take;
            got += take;
        }
        return (uint32_t)v;
    }
    int32_t sbits(int k) {   // two's complement, k <= 32
        if (k == 0) return 0;
        const uint32_t v = bits(k);
        return k == 32 ? (int32_t)v : (int32_t)(v << (32 - k)) >> (32 - k);
    }
    uint32_t unary() {       // number of 0 bits before the next 1 bit
        uint32_t z = 0;
        while (true) {
            if (pos >= n * 8) {
                fail = true;
                return z;
            }
            const int64_t byte = pos >> 3;
            const int off = (int)(pos & 7);
            const uint8_t rest = (uint8_t)(d[byte] << off);
            if (rest) {
                const int lz = __builtin_clz((unsigned)rest) - 24;
                z += lz;
                pos += lz + 1;
                return z;
            }
            z += 8 - off;
            pos += 8 - off;
        }
    }
    void align() { pos = (pos + 7) & ~(int64_t)7; }
};

uint8_t crc8(const uint8_t* p, int64_t n) {          // poly x^8 + x^2 + x + 1, init 0
    uint8_t c = 0;
    for (int64_t i = 0; i < n; ++i) {
        c ^= p[i];
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
    }
    return c;
}
uint16_t crc16(const uint8_t* p, int64_t n) {        // poly x^16 + x^15 + x^2 + 1, init 0
    uint16_t c = 0;
    for (int64_t i = 0; i < n; ++i) {
        c ^= (uint16_t)(p[i] << 8);
        for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
    }
    return c;
}

struct Md5 {                                          // RFC 1321
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t len = 0;
    uint8_t buf[64];
    int fill = 0;
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
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D;
            D = C;
            C = B;
            B = B + rol(A + f + K[i] + m[g], S[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, int64_t n) {
        len += (uint64_t)n;
        while (n > 0) {
            const int take = (64 - fill) < n ? (64 - fill) : (int)n;
            for (int i = 0; i < take; ++i) buf[fill + i] = p[i];
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
        update(l, 8);
        const uint32_t v[4] = {a, b, c, d};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
    }
};

struct FlacInfo {
    int sr, channels, bps, min_block, max_block;
    int64_t total;
    uint8_t md5[16];
    int64_t audio_start;     // byte offset of the first frame
};

int flac_parse_header(const uint8_t* data, int64_t n, FlacInfo* fi) {
    int64_t p = 0;
    if (n >= 10 && data[0] == 'I' && data[1] == 'D' && data[2] == '3')             // an ID3v2 tag some writers put in front
        p = 10 + (((int64_t)(data[6] & 0x7f) << 21) | ((int64_t)(data[7] & 0x7f) << 14) | ((int64_t)(data[8] & 0x7f) << 7) | (data[9] & 0x7f));
    if (p + 4 > n || data[p] != 'f' || data[p + 1] != 'L' || data[p + 2] != 'a' || data[p + 3] != 'C') {
        set_error("flac: no fLaC marker");
        return LLARK_ERR_INVALID;
    }
    p += 4;
    bool have_info = false, last = false;
    while (!last) {
        if (p + 4 > n) { set_error("flac: truncated metadata"); return LLARK_ERR_INVALID; }
        last = (data[p] & 0x80) != 0;
        const int type = data[p] & 0x7f;
        const int64_t len = ((int64_t)data[p + 1] << 16) | ((int64_t)data[p + 2] << 8) | data[p + 3];
        p += 4;
        if (p + len > n) { set_error("flac: truncated metadata block"); return LLARK_ERR_INVALID; }
        if (type == 0) {
            if (len < 34) { set_error("flac: STREAMINFO of %lld bytes", (long long)len); return LLARK_ERR_INVALID; }
            BitReader br{data + p, len, 0, false};
            fi->min_block = (int)br.bits(16);
            fi->max_block = (int)br.bits(16);
            br.bits(24);
            br.bits(24);
            fi->sr = (int)br.bits(20);
            fi->channels = (int)br.bits(3) + 1;
            fi->bps = (int)br.bits(5) + 1;
            fi->total = ((int64_t)br.bits(4) << 32) | br.bits(32);
            for (int i = 0; i < 16; ++i) fi->md5[i] = (uint8_t)br.bits(8);
            have_info = true;
        }
        p += len;
    }
    if (!have_info) { set_error("flac: no STREAMINFO block"); return LLARK_ERR_INVALID; }
    if (fi->sr <= 0 || fi->bps < 4 || fi->bps > 32) { set_error("flac: STREAMINFO says %d Hz, %d bits", fi->sr, fi->bps); return LLARK_ERR_INVALID; }
    fi->audio_start = p;
    return LLARK_OK;
}

// one subframe of `bs` samples at `bps` bits into out[0..bs)
int flac_subframe(BitReader& br, int bs, int bps, int64_t* out) {
    if (br.bits(1)) { set_error("flac: subframe padding bit set"); return LLARK_ERR_INVALID; }
    const int type = (int)br.bits(6);
    int wasted = 0;
    if (br.bits(1)) wasted = (int)br.unary() + 1;
    if (wasted >= bps) { set_error("flac: %d wasted bits of %d", wasted, bps); return LLARK_ERR_INVALID; }
    bps -= wasted;
    if (bps > 33) { set_error("flac: %d-bit subframe", bps); return LLARK_ERR_UNSUPPORTED; }
    auto sample = [&](int k) -> int64_t {                       // k-bit signed, k <= 33 (a 32-bit stream's side channel)
        if (k <= 32) return br.sbits(k);
        const int64_t hi = br.sbits(k - 32);
        return hi * 4294967296ll + (int64_t)br.bits(32);
    };
    int order = 0;
    if (type == 0) {
        const int64_t v = sample(bps);
        for (int i = 0; i < bs; ++i) out[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < bs; ++i) out[i] = sample(bps);
    } else if (type >= 8 && type <= 12) {
        order = type - 8;
    } else if (type >= 32) {
        order = type - 31;
    } else {
        set_error("flac: reserved subframe type %d", type);
        return LLARK_ERR_INVALID;
    }
    if (type >= 8) {
        if (order > bs) { set_error("flac: predictor order %d for a block of %d", order, bs); return LLARK_ERR_INVALID; }
        for (int i = 0; i < order; ++i) out[i] = sample(bps);
        int coef[32], shift = 0;
        if (type >= 32) {
            const int prec = (int)br.bits(4) + 1;
            if (prec == 16) { set_error("flac: invalid LPC precision"); return LLARK_ERR_INVALID; }
            shift = br.sbits(5);
            if (shift < 0) { set_error("flac: negative LPC shift"); return LLARK_ERR_INVALID; }
            for (int i = 0; i < order; ++i) coef[i] = br.sbits(prec);
        } else {
            static const int fixed[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
            for (int i = 0; i < order; ++i) coef[i] = fixed[order][i];
        }
        // residual: partitioned Rice
        const int method = (int)br.bits(2);
        if (method > 1) { set_error("flac: reserved residual coding method %d", method); return LLARK_ERR_INVALID; }
        const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
        const int porder = (int)br.bits(4);
        const int parts = 1 << porder;
        if ((bs >> porder) << porder != bs || (bs >> porder) < order) {
            set_error("flac: block of %d does not split into %d partitions after %d warm-up samples", bs, parts, order);
            return LLARK_ERR_INVALID;
        }
        int i = order;
        for (int pt = 0; pt < parts; ++pt) {
            const int cnt = (bs >> porder) - (pt == 0 ? order : 0);
            const int k = (int)br.bits(pbits);
            if (k == esc) {
                const int raw = (int)br.bits(5);
                for (int j = 0; j < cnt; ++j) out[i++] = br.sbits(raw);
            } else {
                for (int j = 0; j < cnt; ++j) {
                    const uint64_t u = ((uint64_t)br.unary() << k) | (k ? br.bits(k) : 0u);
                    out[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
                }
            }
            if (br.fail) { set_error("flac: truncated residual"); return LLARK_ERR_INVALID; }
        }
        for (int t = order; t < bs; ++t) {
            int64_t pred = 0;
            for (int j = 0; j < order; ++j) pred += (int64_t)coef[j] * out[t - 1 - j];
            out[t] += pred >> shift;
        }
    }
    if (wasted)
        for (int i = 0; i < bs; ++i) out[i] *= (int64_t)1 << wasted;
    if (br.fail) { set_error("flac: truncated subframe"); return LLARK_ERR_INVALID; }
    return LLARK_OK;
}

}  // namespace

// STREAMINFO of a FLAC stream in memory: sample rate, channels, bits per sample, total samples per channel (0 = not recorded).
extern "C" int llark_flac_info_host(const uint8_t* data, int64_t n, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples) {
    LLARK_REQUIRE(data && n > 0 && sample_rate && channels && bits_per_sample && total_samples, "flac_info_host: null pointer or empty input");
    FlacInfo fi;
    const int rc = flac_parse_header(data, n, &fi);
    if (rc != LLARK_OK) return rc;
    *sample_rate = fi.sr; *channels = fi.channels; *bits_per_sample = fi.bps; *total_samples = fi.total;
    return LLARK_OK;
}

// Decodes every frame.  out (may be NULL: count only) receives interleaved int32 samples [frames][channels], at most cap_frames
// frames; *decoded_frames = frames in the stream.  Every frame's CRC-8 / CRC-16 is checked; with verify_md5 the MD5 of the decoded
// samples is compared with STREAMINFO's (skipped when that field is all zero = "not computed").
extern "C" int llark_flac_decode_host(const uint8_t* data, int64_t n, int32_t* out, int64_t cap_frames, int64_t* decoded_frames, int verify_md5) {
    LLARK_REQUIRE(data && n > 0 && decoded_frames, "flac_decode_host: null pointer or empty input");
    FlacInfo fi;
    int rc = flac_parse_header(data, n, &fi);
    if (rc != LLARK_OK) return rc;
    LLARK_REQUIRE(fi.bps <= 32, "flac_decode_host: %d-bit samples", fi.bps);
    const int ch = fi.channels;
    const int bytes_ps = (fi.bps + 7) / 8;
    bool md5_known = false;
    for (int i = 0; i < 16; ++i) md5_known |= fi.md5[i] != 0;
    const bool do_md5 = verify_md5 && md5_known;
    Md5 md5;
    std::vector<int64_t> chan((size_t)ch * 65536);
    std::vector<uint8_t> raw;
    int64_t p = fi.audio_start, frames = 0;
    while (p + 2 <= n) {
        if (!(data[p] == 0xff && (data[p + 1] & 0xfe) == 0xf8)) {
            // trailing padding / tags after the last frame are tolerated once every recorded sample has been decoded
            if (fi.total && frames >= fi.total) break;
            set_error("flac: lost frame sync at byte %lld", (long long)p);
            return LLARK_ERR_INVALID;
        }
        BitReader br{data + p, n - p, 0, false};
        br.bits(15);
        br.bits(1);                                                 // blocking strategy: only changes what the coded number counts
        const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ch_code = (int)br.bits(4), ss_code = (int)br.bits(3);
        if (br.bits(1)) { set_error("flac: reserved header bit set"); return LLARK_ERR_INVALID; }
        {   // UTF-8 style coded frame / sample number (1..7 bytes)
            const uint32_t b0 = br.bits(8);
            int extra = 0;
            if (b0 & 0x80) {
                uint32_t m = 0x40;
                while (b0 & m) { ++extra; m >>= 1; }
                if (extra == 0 || extra > 6) { set_error("flac: bad coded number"); return LLARK_ERR_INVALID; }
            }
            for (int i = 0; i < extra; ++i)
                if ((br.bits(8) & 0xc0) != 0x80) { set_error("flac: bad coded number"); return LLARK_ERR_INVALID; }
        }
        int bs;
        if (bs_code == 0) { set_error("flac: reserved block size code"); return LLARK_ERR_INVALID; }
        else if (bs_code == 1) bs = 192;
        else if (bs_code <= 5) bs = 576 << (bs_code - 2);
        else if (bs_code == 6) bs = (int)br.bits(8) + 1;
        else if (bs_code == 7) bs = (int)br.bits(16) + 1;
        else bs = 256 << (bs_code - 8);
        if (sr_code == 12) br.bits(8);
        else if (sr_code == 13 || sr_code == 14) br.bits(16);
        else if (sr_code == 15) { set_error("flac: invalid sample rate code"); return LLARK_ERR_INVALID; }
        const int64_t hdr_bytes = br.pos >> 3;
        const uint8_t c8 = (uint8_t)br.bits(8);
        if (br.fail || crc8(data + p, hdr_bytes) != c8) { set_error("flac: frame header CRC mismatch at byte %lld", (long long)p); return LLARK_ERR_INVALID; }
        static const int ss_bits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
        const int bps = ss_code == 0 ? fi.bps : ss_bits[ss_code];
        if (bps < 0) { set_error("flac: reserved sample size code"); return LLARK_ERR_INVALID; }
        int nch;
        if (ch_code <= 7) nch = ch_code + 1;
        else if (ch_code <= 10) nch = 2;
        else { set_error("flac: reserved channel assignment %d", ch_code); return LLARK_ERR_INVALID; }
        if (nch != ch || bps != fi.bps) { set_error("flac: frame with %d channels x %d bits in a %d x %d stream", nch, bps, ch, fi.bps); return LLARK_ERR_UNSUPPORTED; }
        for (int c = 0; c < nch; ++c) {
            const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            rc = flac_subframe(br, bs, bps + (side ? 1 : 0), chan.data() + (size_t)c * 65536);
            if (rc != LLARK_OK) return rc;
        }
        br.align();
        const int64_t body = br.pos >> 3;
        const uint16_t c16 = (uint16_t)br.bits(16);
        if (br.fail || crc16(data + p, body) != c16) { set_error("flac: frame CRC mismatch at byte %lld", (long long)p); return LLARK_ERR_INVALID; }
        int64_t* c0 = chan.data();
        int64_t* c1 = chan.data() + 65536;
        if (ch_code == 8) for (int i = 0; i < bs; ++i) c1[i] = c0[i] - c1[i];                 // left, side  -> right = left - side
        else if (ch_code == 9) for (int i = 0; i < bs; ++i) c0[i] = c0[i] + c1[i];            // side, right -> left = right + side
        else if (ch_code == 10)
            for (int i = 0; i < bs; ++i) {                                                     // mid, side
                const int64_t side = c1[i], mid = c0[i] * 2 + (side & 1);
                c0[i] = (mid + side) >> 1;
                c1[i] = (mid - side) >> 1;
            }
        if (do_md5) {
            raw.resize((size_t)bs * nch * bytes_ps);
            size_t o = 0;
            for (int i = 0; i < bs; ++i)
                for (int c = 0; c < nch; ++c) {
                    const int64_t v = chan[(size_t)c * 65536 + i];
                    for (int b = 0; b < bytes_ps; ++b) raw[o++] = (uint8_t)(v >> (8 * b));
                }
            md5.update(raw.data(), (int64_t)raw.size());
        }
        if (out)
            for (int i = 0; i < bs && frames + i < cap_frames; ++i)
                for (int c = 0; c < nch; ++c) out[(size_t)(frames + i) * nch + c] = (int32_t)chan[(size_t)c * 65536 + i];
        frames += bs;
        p += br.pos >> 3;
    }
    if (fi.total && frames != fi.total) { set_error("flac: decoded %lld samples per channel, STREAMINFO says %lld", (long long)frames, (long long)fi.total); return LLARK_ERR_INVALID; }
    if (do_md5) {
        uint8_t got[16];
        md5.finish(got);
        for (int i = 0; i < 16; ++i)
            if (got[i] != fi.md5[i]) { set_error("flac: MD5 of the decoded audio differs from STREAMINFO's signature"); return LLARK_ERR_INVALID; }
    }
    *decoded_frames = frames;
    return LLARK_OK;
}
