// psnd_core.hip - version, error string, integer framing contract (host side).
#include "psnd_common.h"

static thread_local char g_err[512] = "";

void psnd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int psnd_version(void) { return 110; }  // 0.1.10: + conv chain / paired backward, optimizer, loss, data, PQMF entry points

extern "C" const char *psnd_last_error(void) { return g_err; }

static inline int64_t pad_of(int n_fft, int hop, int framing) {
    if (framing == PSND_FRAMING_NONE) return 0;
    return framing == PSND_FRAMING_CENTER ? n_fft / 2 : (n_fft - hop) / 2;
}

// pytorch_sound/models/transforms.py:55-66 (pad n/2 + conv1d stride hop, no padding) and
// :352-360 (pad (n-h)/2 + torch.stft(center=False)).
extern "C" int64_t psnd_frame_count(int64_t T, int n_fft, int hop, int framing) {
    if (T <= 0 || n_fft <= 0 || hop <= 0) return 0;
    const int64_t L = T + 2 * pad_of(n_fft, hop, framing);
    if (L < n_fft) return 0;
    return (L - n_fft) / hop + 1;
}

// F.pad(mode='reflect'): x[-i] = x[i], x[T-1+i] = x[T-1-i]
extern "C" int64_t psnd_frame_sample_index(int64_t f, int m, int64_t T, int n_fft, int hop, int framing) {
    int64_t i = f * hop - pad_of(n_fft, hop, framing) + m;
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}
