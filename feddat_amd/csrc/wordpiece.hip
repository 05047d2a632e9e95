// Text half of the input pipeline: BERT WordPiece tokenisation of a batch of questions on the device,
// once per batch -- what ViltProcessor(text=..., padding=True, truncation=True, max_length=40) (src/modeling/vilt.py:98)
// and BertTokenizer(..., max_length=25) (src/modeling/albef.py:56-57) do on the host, three times per batch.
//
// Byte / integer work, one wave per question: the text is normalised into LDS (ASCII lower-casing, whitespace
// folding), then walked piece by piece; for every piece the 64 lanes test 64 candidate end positions at once (longest
// first), each lane hashing its candidate (FNV-1a 64 + a 32-bit polynomial check hash) and probing an open-addressing
// table of the vocabulary; a ballot picks the longest match.  Unicode work that needs tables (NFD accent stripping,
// CJK / non-ASCII punctuation spacing, non-ASCII lower-casing) is done by the host binding for the rare texts that need
// it (feddat_amd/tokenization.py) -- the kernel treats UTF-8 multi-byte sequences as word characters and only ever cuts a
// word at a character boundary.  A text with raw control characters, or longer than the LDS window, is flagged
// (out_len = -1) instead of being tokenised wrongly.
#include <string.h>

#include "common.hip.h"

namespace {

constexpr int TEXT_MAX = 2048;      // bytes of one normalised text
constexpr int TOK_CAP = 384;        // pieces kept per text (only the first max_len - 2 are ever used)

struct Entry {                      // 16 bytes
    uint64_t key;                   // FNV-1a 64 of the token bytes ("##" prefix included for continuation pieces)
    uint32_t chk;                   // polynomial hash (base 131) + length, second opinion against 64-bit collisions
    int32_t id;                     // -1 = empty slot
};

__host__ __device__ inline void hash_step(uint64_t& h, uint32_t& c, unsigned char b) {
    h = (h ^ b) * 1099511628211ull;
    c = c * 131u + b + 1u;
}

__device__ __forceinline__ bool is_punct_ascii(unsigned char c) {
    return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}

__device__ __forceinline__ int probe(const Entry* __restrict__ tab, uint32_t mask, uint64_t h, uint32_t c) {
    uint32_t i = (uint32_t)h & mask;
    for (;;) {
        const Entry e = tab[i];
        if (e.id < 0) return -1;
        if (e.key == h && e.chk == c) return e.id;
        i = (i + 1) & mask;
    }
}

__global__ __launch_bounds__(64) void wordpiece_encode_kernel(const unsigned char* __restrict__ text,
                                                              const long* __restrict__ offsets,
                                                              const Entry* __restrict__ tab, uint32_t mask, int unk,
                                                              int cls, int sep, int pad, int max_len,
                                                              long* __restrict__ out_ids, long* __restrict__ out_mask,
                                                              int* __restrict__ out_len) {
    __shared__ unsigned char buf[TEXT_MAX];
    __shared__ int tok[TOK_CAP];
    const int t = blockIdx.x, lane = threadIdx.x;
    const long beg = offsets[t];
    const int len = (int)(offsets[t + 1] - beg);
    long* ids_row = out_ids + (size_t)t * max_len;
    long* mask_row = out_mask + (size_t)t * max_len;
    bool bad = len > TEXT_MAX;
    for (int i = lane; i < len && i < TEXT_MAX; i += 64) {
        unsigned char c = text[beg + i];
        if (c == 9 || c == 10 || c == 13) c = ' ';
        else if (c < 32 || c == 127) bad = true;          // raw control characters: the host normaliser drops those
        else if (c >= 'A' && c <= 'Z') c += 32;
        buf[i] = c;
    }
    if (__ballot(bad)) {
        if (lane == 0) out_len[t] = -1;
        return;
    }
    __syncthreads();
    const int body = max_len - 2;
    int pos = 0, ntok = 0;                                  // wave-uniform
    while (pos < len && ntok < body) {
        const unsigned char c0 = buf[pos];
        if (c0 == ' ') { ++pos; continue; }
        if (is_punct_ascii(c0)) {                           // punctuation is a token of its own
            uint64_t h = 14695981039346656037ull;
            uint32_t c = 0;
            hash_step(h, c, c0);
            const int id = probe(tab, mask, h, c + 1u * 0x9e3779b1u);
            if (lane == 0) tok[ntok] = id < 0 ? unk : id;
            ++ntok;
            ++pos;
            continue;
        }
        int end = pos + 1, nchar = 1;
        while (end < len && buf[end] != ' ' && !is_punct_ascii(buf[end])) {
            nchar += (buf[end] & 0xC0) != 0x80;             // UTF-8 lead bytes count characters
            ++end;
        }
        const int wlen = end - pos;
        const int tok0 = ntok;
        bool unk_word = nchar > 100;                        // max_input_chars_per_word
        int start = 0;
        while (!unk_word && start < wlen) {
            const int rem = wlen - start;
            int found_len = 0, found_id = -1;
            for (int base = 0; base < rem && !found_len; base += 64) {
                const int L = rem - base - lane;            // this lane's candidate length, longest first
                int id = -1;
                if (L >= 1 && (start + L == wlen || (buf[pos + start + L] & 0xC0) != 0x80)) {
                    uint64_t h = 14695981039346656037ull;
                    uint32_t c = 0;
                    int n = L;
                    if (start > 0) {
                        hash_step(h, c, '#');
                        hash_step(h, c, '#');
                        n += 2;
                    }
                    for (int k = 0; k < L; ++k) hash_step(h, c, buf[pos + start + k]);
                    id = probe(tab, mask, h, c + (uint32_t)n * 0x9e3779b1u);
                }
                const unsigned long long hit = __ballot(id >= 0);
                if (hit) {
                    const int first = __builtin_ctzll(hit);
                    found_len = rem - base - first;
                    found_id = __shfl(id, first, 64);
                }
            }
            if (!found_len) { unk_word = true; break; }
            if (ntok < TOK_CAP) {
                if (lane == 0) tok[ntok] = found_id;
                ++ntok;
            }
            start += found_len;
        }
        if (unk_word) {                                     // the whole word becomes [UNK]
            ntok = tok0;
            if (lane == 0) tok[ntok] = unk;
            ++ntok;
        }
        pos = end;
    }
    __syncthreads();
    const int nb = ntok < body ? ntok : body;
    for (int i = lane; i < max_len; i += 64) {
        long v = pad;
        if (i == 0) v = cls;
        else if (i <= nb) v = tok[i - 1];
        else if (i == nb + 1) v = sep;
        ids_row[i] = v;
        mask_row[i] = i <= nb + 1 ? 1 : 0;
    }
    if (lane == 0) out_len[t] = nb + 2;
}

}  // namespace

extern "C" long feddat_wordpiece_table_entries(int n_vocab) {
    long n = 64;
    while (n < 2L * n_vocab) n <<= 1;
    return n;
}

// HOST function: builds the open-addressing table (16 bytes per entry) of a vocabulary given as a blob of token bytes
// (token i = blob[offsets[i] .. offsets[i+1])) into host memory; the caller uploads it.
extern "C" int feddat_wordpiece_table_build(const char* blob, const long* offsets, int n_vocab, void* table_host,
                                            long entries) {
    FD_CHECK_ARG(blob && offsets && table_host && n_vocab > 0 && entries >= 2L * n_vocab && (entries & (entries - 1)) == 0);
    Entry* tab = (Entry*)table_host;
    for (long i = 0; i < entries; ++i) tab[i] = Entry{0, 0, -1};
    const uint32_t mask = (uint32_t)(entries - 1);
    for (int v = 0; v < n_vocab; ++v) {
        uint64_t h = 14695981039346656037ull;
        uint32_t c = 0;
        const long n = offsets[v + 1] - offsets[v];
        if (n <= 0) continue;
        for (long k = 0; k < n; ++k) hash_step(h, c, (unsigned char)blob[offsets[v] + k]);
        c += (uint32_t)n * 0x9e3779b1u;
        uint32_t i = (uint32_t)h & mask;
        bool dup = false;
        while (tab[i].id >= 0) {
            if (tab[i].key == h && tab[i].chk == c) { dup = true; break; }      // first occurrence wins
            i = (i + 1) & mask;
        }
        if (!dup) tab[i] = Entry{h, c, v};
    }
    return FEDDAT_OK;
}

extern "C" int feddat_wordpiece_encode(const void* text, const long* offsets, int n_texts, const void* table,
                                       long entries, int unk_id, int cls_id, int sep_id, int pad_id, int max_len,
                                       long* out_ids, long* out_mask, int* out_len, hipStream_t stream) {
    FD_CHECK_ARG(text && offsets && table && n_texts > 0 && entries > 0 && (entries & (entries - 1)) == 0);
    FD_CHECK_ARG(out_ids && out_mask && out_len && max_len >= 3 && max_len <= TOK_CAP);
    hipLaunchKernelGGL(wordpiece_encode_kernel, dim3(n_texts), dim3(64), 0, stream, (const unsigned char*)text, offsets,
                       (const Entry*)table, (uint32_t)(entries - 1), unk_id, cls_id, sep_id, pad_id, max_len, out_ids,
                       out_mask, out_len);
    FD_LAUNCH_RET();
}
