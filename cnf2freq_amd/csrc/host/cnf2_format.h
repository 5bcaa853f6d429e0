// cnf2_format.h -- the characters printf would write, without printf.
//
// The output of a run is text: three numbers per (individual, marker) for the per-locus rows (cnF2freq.cpp:6183-6188) and
// eleven for the dump of the haplotypes (cnF2freq.cpp:8157-8192) -- for 25 000 individuals x 10 000 markers half a billion
// lines.  fprintf takes 0.36 us for a row line and 0.92 us for a dump line, which is five minutes of a run whose hundred
// iterations take two.  Here a number is formatted by one multiplication and an integer division, and ONLY where that is
// provably what printf prints: "%.Nf" rounds the exact binary value to N decimals, to nearest, ties to even; the product
// a * 10^N carries one rounding of relative size 2^-53, so whenever its fractional part is further than that from 1/2 the
// integer nearest to it is the integer printf's digits spell.  Anything else (a tie within the error bound, 1e9 and
// above, infinities, NaN) goes to snprintf.  tests/test_host_format.py compares millions of values, the adversarial ones
// (k + 1/2) 10^-N +- a few ulp included, with snprintf character by character.
#ifndef CNF2_FORMAT_H
#define CNF2_FORMAT_H

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

namespace cnf2host {

// appends |v| in decimal (at least one digit); returns the new end
inline char* fmt_uint(char* p, uint64_t v)
{
    char tmp[24];
    int  n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline char* fmt_int(char* p, int v)
{
    if (v < 0) {
        *p++ = '-';
        return fmt_uint(p, (uint64_t)(-(int64_t)v));
    }
    return fmt_uint(p, (uint64_t)v);
}

// appends d as printf("%.<decimals>f") writes it (decimals 1..9; at most 32 characters); returns the new end
inline char* fmt_fixed(char* p, double d, int decimals)
{
    static const double   P10[10] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
    static const uint64_t I10[10] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull};
    const double a = fabs(d);
    if (a < 1e9) {                                         // (false for NaN)
        const double s    = a * P10[decimals];             // one rounding: |s - a 10^N| <= 2^-53 s
        const double fl   = floor(s);
        const double frac = s - fl;                        // exact
        const double off  = fabs(frac - 0.5);
        if (s < 4.0e15 && off > s * 2.3e-16 + 1e-300) {    // below 2^52: floor and frac are exact; 2.3e-16 > 2^-52, twice the bound
            uint64_t q = (uint64_t)fl + (frac > 0.5 ? 1 : 0);
            if (signbit(d)) *p++ = '-';
            p    = fmt_uint(p, q / I10[decimals]);
            *p++ = '.';
            uint64_t f = q % I10[decimals];
            for (int k = decimals - 1; k >= 0; k--) {
                p[k] = (char)('0' + f % 10);
                f /= 10;
            }
            return p + decimals;
        }
    }
    return p + snprintf(p, 400, "%.*f", decimals, d);       // (a double below 1e9 or not finite: well under 400 characters)
}

// ---- reading the dump back (Engine::deserialize): a marker's line is read by sscanf(line, "%lf %d %d %lf %lf %lf") in the
// reference (cnF2freq.cpp:7994-8062), 0.6 us a line.  The numbers a dump holds are plain decimals of a few digits: such a
// number is (integer of its digits) / 10^(digits behind the point), both exact in a double when the integer is below 2^53
// and the power at most 10^22, and one IEEE division rounds correctly -- which is what strtod returns.  Anything else in
// a line (exponents, "nan", hexadecimal floats, more than 18 digits, a field that does not end at a blank) sends the whole
// line to sscanf.
inline bool scan_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool fast_decimal(const char*& p, double* out)
{
    static const double P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    while (scan_space(*p)) p++;
    bool neg = false;
    if (*p == '-') {
        neg = true;
        p++;
    } else if (*p == '+') p++;
    uint64_t v = 0;
    int      nd = 0, nf = 0;
    while (*p >= '0' && *p <= '9') {
        v = v * 10 + (uint64_t)(*p++ - '0');
        if (++nd > 18) return false;
    }
    if (*p == '.') {
        p++;
        while (*p >= '0' && *p <= '9') {
            v = v * 10 + (uint64_t)(*p++ - '0');
            nf++;
            if (++nd > 18) return false;
        }
    }
    if (nd == 0 || (*p && !scan_space(*p)) || v >= (1ull << 53)) return false;
    const double r = nf ? (double)v / P10[nf] : (double)v;
    *out = neg ? -r : r;
    return true;
}
inline bool fast_int(const char*& p, int* out)
{
    while (scan_space(*p)) p++;
    bool neg = false;
    if (*p == '-') {
        neg = true;
        p++;
    } else if (*p == '+') p++;
    int64_t v = 0;
    int     nd = 0;
    while (*p >= '0' && *p <= '9') {
        v = v * 10 + (*p++ - '0');
        if (++nd > 9) return false;
    }
    if (nd == 0 || (*p && !scan_space(*p))) return false;
    *out = (int)(neg ? -v : v);
    return true;
}
// the fields of a dump line; returns what sscanf(line, "%lf %d %d %lf %lf %lf", ...) returns
inline int parse_dump_line(const char* line, double* hw, int* a, int* b, double* negshift, double* s1, double* s2)
{
    const char* p = line;
    double      f[4];
    int         i[2];
    if (fast_decimal(p, &f[0]) && fast_int(p, &i[0]) && fast_int(p, &i[1]) && fast_decimal(p, &f[1]) && fast_decimal(p, &f[2]) &&
        fast_decimal(p, &f[3])) {
        *hw = f[0];
        *a = i[0];
        *b = i[1];
        *negshift = f[1];
        *s1 = f[2];
        *s2 = f[3];
        return 6;
    }
    return sscanf(line, "%lf %d %d %lf %lf %lf", hw, a, b, negshift, s1, s2);
}

// a growing character buffer with room guaranteed before every number
struct TextBuf {
    std::string s;
    size_t      n = 0;
    void room(size_t more)
    {
        if (n + more > s.size()) s.resize((n + more) * 2 + 4096);
    }
    char* at() { return &s[n]; }
    void  took(char* end) { n = (size_t)(end - s.data()); }
    void  put(char c)
    {
        room(1);
        s[n++] = c;
    }
    void put(const char* t, size_t len)
    {
        room(len);
        memcpy(&s[n], t, len);
        n += len;
    }
    void fixed(double d, int decimals)
    {
        room(420);
        took(fmt_fixed(at(), d, decimals));
    }
    void integer(int v)
    {
        room(16);
        took(fmt_int(at(), v));
    }
    void clear() { n = 0; }
};

}  // namespace cnf2host
#endif
