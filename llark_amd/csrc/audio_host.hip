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
