// Host side of row a6: CLIP's byte-level BPE tokenizer in C++ (no device code in this file; it lives in liblpi_hip.so so that the
// plugin has one native library).  Own implementation of the published scheme, behaviour-checked against ids captured from the
// reference's SimpleTokenizer / clip.tokenize (models/clip/simple_tokenizer.py:62-132, models/clip/clip.py:185-221) and fuzzed
// against the Python implementation in lpi_amd/retrieval/models/clip/simple_tokenizer.py.
//
// Division of labour with the Python caller: text CLEANING (ftfy / html.unescape / whitespace collapse / lower-casing — Python's
// own Unicode machinery) stays in Python; this file does the expensive part: pattern split, byte mapping, pair merging, ids.
#include <stdint.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lpi_hip.h"
#include "unicode_ln.h"

namespace {

constexpr int N_MERGES = 49152 - 256 - 2;     // merges CLIP uses (simple_tokenizer.py:66)
const char* const SOT_TEXT = "<|startoftext|>";
const char* const EOT_TEXT = "<|endoftext|>";

bool in_ranges(const unsigned (*r)[2], int n, unsigned cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < r[mid][0]) hi = mid - 1;
        else if (cp > r[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
bool is_letter(unsigned cp) { return in_ranges(kUnicodeLetters, kUnicodeLetters_n, cp); }
bool is_number(unsigned cp) { return in_ranges(kUnicodeNumbers, kUnicodeNumbers_n, cp); }
// \s of the `regex` module (the caller has already collapsed whitespace runs to ' ', so this is belt and braces)
bool is_space(unsigned cp) {
    return (cp >= 0x09 && cp <= 0x0D) || (cp >= 0x1C && cp <= 0x20) || cp == 0x85 || cp == 0xA0 || cp == 0x1680 ||
           (cp >= 0x2000 && cp <= 0x200A) || cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}

// decode one UTF-8 code point at s[i]; returns its byte length (malformed bytes are taken one at a time as U+FFFD-like "other")
int decode(const std::string& s, size_t i, unsigned& cp) {
    const unsigned char c = (unsigned char)s[i];
    auto cont = [&](size_t k) { return i + k < s.size() && ((unsigned char)s[i + k] & 0xC0) == 0x80; };
    if (c < 0x80) { cp = c; return 1; }
    if ((c >> 5) == 6 && cont(1)) { cp = ((c & 0x1F) << 6) | (s[i + 1] & 0x3F); return 2; }
    if ((c >> 4) == 14 && cont(1) && cont(2)) { cp = ((c & 0x0F) << 12) | ((s[i + 1] & 0x3F) << 6) | (s[i + 2] & 0x3F); return 3; }
    if ((c >> 3) == 30 && cont(1) && cont(2) && cont(3)) {
        cp = ((c & 0x07) << 18) | ((s[i + 1] & 0x3F) << 12) | ((s[i + 2] & 0x3F) << 6) | (s[i + 3] & 0x3F);
        return 4;
    }
    cp = 0xFFFD;
    return 1;
}

void append_utf8(std::string& out, unsigned cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
}

struct Bpe {
    std::string byte_sym[256];                                   // byte -> its printable stand-in (UTF-8)
    std::unordered_map<std::string, int> encoder;                // symbol -> id
    std::unordered_map<std::string, int> rank;                   // "left right" -> merge priority
    std::unordered_map<std::string, std::vector<int>> memo;      // pre-token -> ids
    int sot = 0, eot = 0;

    bool init(const char* text, long n) {
        // byte alphabet (bytes_to_unicode): printable latin-1 bytes map to themselves, the others to U+0100.. in byte order;
        // vocabulary order: the printable ones first, then the others — each once plain, once with the word-end marker
        std::vector<int> printable, rest;
        for (int b = 0; b < 256; ++b) {
            const bool keep = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
            (keep ? printable : rest).push_back(b);
        }
        int extra = 0;
        for (int b : printable) append_utf8(byte_sym[b], (unsigned)b);
        for (int b : rest) append_utf8(byte_sym[b], 256u + (unsigned)extra++);
        std::vector<std::string> symbols;
        for (int b : printable) symbols.push_back(byte_sym[b]);
        for (int b : rest) symbols.push_back(byte_sym[b]);
        const size_t nb = symbols.size();
        for (size_t i = 0; i < nb; ++i) symbols.push_back(symbols[i] + "</w>");
        // merge table: first line is a header; then N_MERGES lines "left right"
        const char* p = text;
        const char* end = text + n;
        auto next_line = [&](std::string& line) {
            if (p >= end) return false;
            const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!q) q = end;
            line.assign(p, (size_t)(q - p));
            if (!line.empty() && line.back() == '\r') line.pop_back();
            p = q < end ? q + 1 : end;
            return true;
        };
        std::string line;
        if (!next_line(line)) return false;
        for (int m = 0; m < N_MERGES; ++m) {
            if (!next_line(line)) return false;
            const size_t sp = line.find(' ');
            if (sp == std::string::npos || sp == 0 || sp + 1 >= line.size()) return false;
            rank.emplace(line, m);
            symbols.push_back(line.substr(0, sp) + line.substr(sp + 1));
        }
        symbols.push_back(SOT_TEXT);
        symbols.push_back(EOT_TEXT);
        for (size_t i = 0; i < symbols.size(); ++i) encoder.emplace(symbols[i], (int)i);   // first occurrence wins, as dict(zip()) would not matter: symbols are unique
        sot = encoder[SOT_TEXT];
        eot = encoder[EOT_TEXT];
        return true;
    }

    // greedy lowest-rank-first pair merging of one pre-token
    const std::vector<int>& merge_word(const std::string& tok) {
        auto it = memo.find(tok);
        if (it != memo.end()) return it->second;
        std::vector<std::string> parts;
        for (unsigned char c : tok) parts.push_back(byte_sym[c]);
        std::vector<int> ids;
        if (!parts.empty()) {
            parts.back() += "</w>";
            while (parts.size() > 1) {
                int best = -1;
                size_t where = 0;
                std::string key;
                for (size_t i = 0; i + 1 < parts.size(); ++i) {
                    key.assign(parts[i]);
                    key += ' ';
                    key += parts[i + 1];
                    auto r = rank.find(key);
                    if (r != rank.end() && (best < 0 || r->second < best)) { best = r->second; where = i; }
                }
                if (best < 0) break;
                const std::string a = parts[where], b = parts[where + 1];
                std::vector<std::string> merged;
                for (size_t i = 0; i < parts.size();) {
                    if (i + 1 < parts.size() && parts[i] == a && parts[i + 1] == b) { merged.push_back(a + b); i += 2; }
                    else { merged.push_back(parts[i]); i += 1; }
                }
                parts.swap(merged);
            }
            for (const auto& s : parts) {
                auto e = encoder.find(s);
                ids.push_back(e == encoder.end() ? 0 : e->second);
            }
        }
        return memo.emplace(tok, std::move(ids)).first->second;
    }

    // <|startoftext|> | <|endoftext|> | 's | 't | 're | 've | 'm | 'll | 'd | [\p{L}]+ | [\p{N}] | [^\s\p{L}\p{N}]+   on cleaned, lower-cased text
    void encode(const std::string& s, std::vector<int>& out) {
        static const char* const contractions[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
        size_t i = 0;
        const size_t n = s.size();
        while (i < n) {
            if (s.compare(i, strlen(SOT_TEXT), SOT_TEXT) == 0) { out.push_back(sot); i += strlen(SOT_TEXT); continue; }
            if (s.compare(i, strlen(EOT_TEXT), EOT_TEXT) == 0) { out.push_back(eot); i += strlen(EOT_TEXT); continue; }
            bool hit = false;
            if (s[i] == '\'') {
                for (const char* c : contractions) {
                    const size_t l = strlen(c);
                    if (s.compare(i, l, c) == 0) {
                        const auto& ids = merge_word(s.substr(i, l));
                        out.insert(out.end(), ids.begin(), ids.end());
                        i += l;
                        hit = true;
                        break;
                    }
                }
            }
            if (hit) continue;
            unsigned cp;
            int len = decode(s, i, cp);
            size_t j = i + len;
            if (is_letter(cp)) {
                while (j < n) {
                    unsigned c2;
                    const int l2 = decode(s, j, c2);
                    if (!is_letter(c2)) break;
                    j += l2;
                }
            } else if (is_number(cp)) {
                // one code point
            } else if (is_space(cp)) {
                i = j;
                continue;
            } else {
                while (j < n) {
                    unsigned c2;
                    const int l2 = decode(s, j, c2);
                    if (is_space(c2) || is_letter(c2) || is_number(c2)) break;
                    j += l2;
                }
            }
            const auto& ids = merge_word(s.substr(i, j - i));
            out.insert(out.end(), ids.begin(), ids.end());
            i = j;
        }
    }
};

}  // namespace

extern "C" void* lpi_bpe_create(const char* merges_utf8, long nbytes) {
    if (!merges_utf8 || nbytes <= 0) return nullptr;
    Bpe* b = new Bpe();
    if (!b->init(merges_utf8, nbytes)) {
        delete b;
        return nullptr;
    }
    return b;
}

extern "C" void lpi_bpe_destroy(void* h) { delete (Bpe*)h; }

extern "C" int lpi_bpe_encode(void* h, const char* text_utf8, int32_t* ids, int max_ids) {
    if (!h || !text_utf8 || (max_ids > 0 && !ids) || max_ids < 0) return LPI_EINVAL;
    std::vector<int> out;
    ((Bpe*)h)->encode(text_utf8, out);
    const int n = (int)out.size();
    for (int i = 0; i < n && i < max_ids; ++i) ids[i] = out[i];
    return n;      // the caller compares with max_ids
}

extern "C" int lpi_bpe_tokenize(void* h, const char* const* texts, int n, int context_length, int truncate, int64_t* out) {
    if (!h || !texts || !out || n < 0 || context_length < 2) return LPI_EINVAL;
    Bpe* b = (Bpe*)h;
    std::vector<int> ids;
    for (int t = 0; t < n; ++t) {
        ids.clear();
        if (!texts[t]) return LPI_EINVAL;
        b->encode(texts[t], ids);
        int64_t* row = out + (size_t)t * context_length;
        for (int i = 0; i < context_length; ++i) row[i] = 0;
        int m = (int)ids.size();
        if (m + 2 > context_length) {
            if (!truncate) return t + 1;       // clip.py:218 raises for this text
            m = context_length - 2;
        }
        row[0] = b->sot;
        for (int i = 0; i < m; ++i) row[1 + i] = ids[i];
        row[1 + m] = b->eot;
    }
    return 0;
}
