// cnf2_readers.cpp -- see cnf2_readers.h.
#include "cnf2_readers.h"

#include <math.h>
#include <string.h>

namespace cnf2host {

// getind(string) / getind(int, true): cnF2freq.cpp:6480-6491 and 2456-2514.
// The very first lookup in the reference is getind("0") at static-init time, which maps "0" to a
// null individual; real individuals are numbered 1, 2, ... in order of first mention.
int Pedigree::getind(const std::string& name)
{
    if (index.empty()) index["0"] = -1;
    auto it = index.find(name);
    if (it != index.end()) return it->second;
    Individual ind;
    ind.n    = (int)index.size();          // origindex = indmap.size() (cnF2freq.cpp:6485)
    ind.name = name;
    const size_t M = pos.size();
    ind.allele.assign(M * 2, 0);           // UnknownMarkerVal (cnF2freq.cpp:2488)
    ind.sure.assign(M * 2, 0.0);           // cnF2freq.cpp:2493
    ind.hw.assign(M, 0.5);                 // cnF2freq.cpp:2491
    inds.push_back(ind);
    const int idx = (int)inds.size() - 1;
    index[name]   = idx;
    char buf[64];
    snprintf(buf, sizeof(buf), "Creating %d", ind.n);   // cnF2freq.cpp:2464
    log.push_back(buf);
    return idx;
}

// cnF2freq.cpp:6669-6685
bool read_alpha_map(FILE* in, Pedigree& P)
{
    if (!in) return false;
    double prevval = 1e30, value;
    int    nowmarker = 0;
    while (fscanf(in, "%lf", &value) == 1) {
        if (value < prevval) P.chromstarts.push_back(nowmarker);
        nowmarker++;
        P.pos.push_back(value);
        prevval = value;
    }
    P.chromstarts.push_back((int32_t)P.pos.size());
    return !P.pos.empty();
}

// cnF2freq.cpp:6495-6540
bool read_alpha_ped(FILE* in, Pedigree& P)
{
    if (!in) return false;
    char me[255], father[255], mother[255];
    while (fscanf(in, "%254s %254s %254s", me, father, mother) == 3) {
        char line[255];
        line[0] = 0;
        if (!fgets(line, 255, in)) line[0] = 0;
        int gen = 0;
        if (sscanf(line, "%d", &gen) != 1) gen = 0;
        const int ime = P.getind(me);
        const int iff = P.getind(father);
        const int ifm = P.getind(mother);
        if (ime < 0) continue;                          // an individual called "0": the reference dereferences null
        P.inds[ime].empty = true;                       // cnF2freq.cpp:6511-6513
        if (iff >= 0) P.inds[iff].empty = true;
        if (ifm >= 0) P.inds[ifm].empty = true;
        // The reference reads ifounderf->gen through a null pointer when a gen >= 2 line names "0" as a
        // parent (cnF2freq.cpp:6515); here a missing parent counts as generation 0.
        const int gf = iff >= 0 ? P.inds[iff].gen : 0;
        const int gm = ifm >= 0 ? P.inds[ifm].gen : 0;
        if (gen >= 2 && !gf && !gm) {                   // implicit private F1 parents (cnF2freq.cpp:6515-6527)
            const int rf = P.getind(std::string(me) + "_aux_realf");
            const int rm = P.getind(std::string(me) + "_aux_realm");
            const int real[2] = {rf, rm};
            for (int k = 0; k < 2; k++) {
                P.inds[ime].pars[k]      = real[k];
                P.inds[real[k]].gen      = 1;
                P.inds[real[k]].pars[0]  = iff;
                P.inds[real[k]].pars[1]  = ifm;
                P.inds[real[k]].empty    = true;
            }
            P.inds[ime].gen = gen;
        } else {
            P.inds[ime].gen     = gen;
            P.inds[ime].pars[0] = iff;
            P.inds[ime].pars[1] = ifm;
        }
        if (gen >= 2) P.dous.push_back(ime);            // cnF2freq.cpp:6535-6538
    }
    return true;
}

// pdf of Binomial(n, 1/2) at k: what boost::math::pdf(binomial_distribution<double>(n), k) returns
// for the default success fraction 0.5 (cnF2freq.cpp:6599-6611).
static double binom_half_pdf(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    return exp(lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0) - n * 0.69314718055994530942);
}

// The tokens of a genotype file, as fscanf("%254s") delivers them (runs of non-whitespace, cut after 254 characters), read
// through one large buffer: a genotype file of 25000 individuals x 10000 markers is 250M tokens, and a libc call (or two)
// per token is half a minute of a run whose 100 iterations take two.
struct TokenReader {
    FILE*             in;
    std::vector<char> buf;
    size_t            at = 0, end = 0;
    bool              eof = false;
    explicit TokenReader(FILE* f) : in(f), buf((size_t)4 << 20) {}
    static bool space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }       // isspace() of the C locale
    bool refill()                                                                    // keeps [at, end), appends to it
    {
        if (eof) return false;
        if (at) {
            memmove(buf.data(), buf.data() + at, end - at);
            end -= at;
            at = 0;
        }
        const size_t got = fread(buf.data() + end, 1, buf.size() - end, in);
        end += got;
        if (!got) eof = true;
        return got != 0;
    }
    // the next token: [tok, tok + n), valid until the next call; false at the end of the file
    bool next(const char*& tok, size_t& n)
    {
        for (;;) {
            while (at < end && space(buf[at])) at++;
            if (at < end || !refill()) break;
        }
        if (at >= end) return false;
        size_t e = at;
        for (;;) {
            while (e < end && e - at < 254 && !space(buf[e])) e++;
            if (e < end || e - at >= 254) break;
            const size_t len = e - at;                                               // the token runs into the buffer's end
            const bool   more = refill();                                            // moves the token to the front
            e = at + len;
            if (!more) break;
        }
        tok = buf.data() + at;
        n = e - at;
        at = e;
        return true;
    }
};

// cnF2freq.cpp:6542-6667
bool read_alpha_gen(FILE* in, Pedigree& P)
{
    if (!in) return false;
    const size_t M = P.pos.size();
    const int haplo = P.getind("haplo");                // cnF2freq.cpp:6544-6549
    for (size_t x = 0; x < M; x++) {
        P.inds[haplo].allele[x * 2] = P.inds[haplo].allele[x * 2 + 1] = 9;
        P.inds[haplo].sure[x * 2] = P.inds[haplo].sure[x * 2 + 1] = 0.0;
    }
    TokenReader  tokens(in);
    const char*  tok;
    size_t       toklen;
    while (tokens.next(tok, toklen)) {
        const int ime = P.getind(std::string(tok, toklen));
        if (ime < 0) return false;
        const bool doublehaplo = (P.inds[ime].pars[1] == P.getind("haplo"));
        P.inds[ime].empty = false;
        int data = 0;
        for (size_t x = 0; x < M; x++) {
            char datastr[255];
            int  data2 = 0, numread;
            if (!tokens.next(tok, toklen)) toklen = 0;
            if (toklen == 1 && tok[0] >= '0' && tok[0] <= '9') {                      // what nearly every token is
                data = tok[0] - '0';
                numread = 1;
            } else {
                memcpy(datastr, tok, toklen);
                datastr[toklen] = 0;
                numread = sscanf(datastr, "%d/%d", &data, &data2);
            }
            Individual& I = P.inds[ime];
            I.hw[x] = 0.5;
            if (numread == 1) {                         // cnF2freq.cpp:6565-6588
                uint8_t a0 = 0, a1 = 0;
                switch (data) {
                case 0: a0 = 1; a1 = 1; break;
                case 1: a0 = 1; a1 = 2; break;
                case 2: a0 = 2; a1 = 2; break;
                default: break;
                }
                I.allele[x * 2] = a0;
                I.allele[x * 2 + 1] = a1;
                if (a0 != 0) I.sure[x * 2] = I.sure[x * 2 + 1] = 0.02;
            } else {                                    // read counts a/b (cnF2freq.cpp:6589-6662)
                char buf[300];
                snprintf(buf, sizeof(buf), "%s %d", datastr, numread);
                P.log.push_back(buf);
                if (data == data2 && !data) {
                    I.allele[x * 2] = I.allele[x * 2 + 1] = 0;
                    I.sure[x * 2] = I.sure[x * 2 + 1] = 0.0;
                } else {
                    // Posterior mean of the two per-allele read fractions given the counts (cnF2freq.cpp:6596-6641).
                    // A split (k1, k2) says how many of the `data` allele-1 reads and of the `data2` allele-2 reads
                    // came from the first chromosome; splits are weighted by Binomial(n, 1/2) per allele (tabulated
                    // once) times the likelihood of the reads under the split's own fractions.  Each split is looked
                    // at from the side on which the first chromosome is the allele-1-richer one.
                    const int n1 = data, n2 = data2, nall = n1 + n2;
                    std::vector<double> w1(n1 + 1, 1.0), w2(n2 + 1, 1.0);
                    if (n1) for (int k = 0; k <= n1; k++) w1[k] = binom_half_pdf(n1, k);
                    if (n2) for (int k = 0; k <= n2; k++) w2[k] = binom_half_pdf(n2, k);
                    struct Frac { double first, second; };
                    auto fractions = [&](int k1, int k2) {
                        Frac f = {0.5, 0.5};
                        if (k1 + k2) f.first = k1 / (double)(k1 + k2);
                        if (nall - k1 - k2) f.second = (n2 - k2) / (double)(nall - k1 - k2);
                        return f;
                    };
                    double acc1 = 0, acc2 = 0, total = 0;
                    for (int s1 = 0; s1 <= n1; s1++)
                        for (int s2 = 0; s2 <= n2; s2++) {
                            int  k1 = s1, k2 = s2;
                            Frac f = fractions(k1, k2);
                            if (!(f.first + 1e-9 > 1 - f.second)) {      // mirror the split (always settles in one step)
                                k1 = n1 - k1;
                                k2 = n2 - k2;
                                f  = fractions(k1, k2);
                            }
                            const double like = pow(f.first, k1) * pow(1 - f.first, k2) * pow(f.second, n2 - k2) *
                                                pow(1 - f.second, n1 - k1);
                            const double wt = w1[s1] * w2[s2] * like;
                            acc1 += f.first * wt;
                            acc2 += f.second * wt;
                            total += wt;
                        }
                    const double sure1 = acc1 / total, sure2 = acc2 / total;
                    uint8_t mk[2] = {2, 1};
                    double  ms[2] = {sure1, sure2};
                    for (int k = 0; k < 2; k++)
                        if (ms[k] > 0.5) {                // cnF2freq.cpp:6644-6652
                            ms[k] = 1 - ms[k];
                            mk[k] = (uint8_t)(k + 1);
                        }
                    I.allele[x * 2] = mk[0];
                    I.allele[x * 2 + 1] = mk[1];
                    I.sure[x * 2] = ms[0];
                    I.sure[x * 2 + 1] = ms[1];
                    snprintf(buf, sizeof(buf), "%d/%d turned into %d %d with %lf;%lf", data, data2, mk[0], mk[1], ms[0], ms[1]);
                    P.log.push_back(buf);
                }
                if (doublehaplo) I.allele[x * 2 + 1] = 9;
            }
        }
        Individual& I  = P.inds[ime];
        I.prior_allele = I.allele;                      // cnF2freq.cpp:6664-6665
        I.prior_sure   = I.sure;
        I.has_prior    = true;
    }
    return true;
}

void cap_markers(Pedigree& P, int cap)
{
    if (cap <= 0 || cap >= (int)P.pos.size()) return;
    P.pos.resize(cap);
    std::vector<int32_t> cs;
    for (size_t c = 0; c + 1 < P.chromstarts.size(); c++)
        if (P.chromstarts[c] < cap) cs.push_back(P.chromstarts[c]);
    cs.push_back(cap);
    P.chromstarts = cs;
    for (auto& I : P.inds) {
        I.allele.resize((size_t)cap * 2);
        I.sure.resize((size_t)cap * 2);
        I.hw.resize(cap);
        if (I.has_prior) {
            I.prior_allele.resize((size_t)cap * 2);
            I.prior_sure.resize((size_t)cap * 2);
        }
    }
}

void build_tables(const Pedigree& P, Tables& T, bool share_blank)
{
    const size_t M = P.pos.size();
    const int    R = (int)P.inds.size();
    T.par.assign((size_t)R * 2, -1);
    T.gen.assign(R, 0);
    T.empty.assign(R, 0);
    T.row_of.assign(R, 0);
    T.dous.assign(P.dous.begin(), P.dous.end());
    T.allele.assign(M * 2, 0);
    T.sure.assign(M * 2, 0.0);
    T.hw.assign(M, 0.5);
    T.n_rows = 1;
    if (!share_blank) {                      // every individual gets a row: no reallocation while they are appended
        T.allele.reserve(((size_t)R + 1) * M * 2);
        T.sure.reserve(((size_t)R + 1) * M * 2);
        T.hw.reserve(((size_t)R + 1) * M);
    }
    for (int r = 0; r < R; r++) {
        const Individual& I = P.inds[r];
        T.par[r * 2]     = I.pars[0];
        T.par[r * 2 + 1] = I.pars[1];
        T.gen[r]         = I.gen;
        T.empty[r]       = I.empty ? 1 : 0;
        bool blank = true;
        for (size_t x = 0; x < M && blank; x++)
            if (I.allele[x * 2] || I.allele[x * 2 + 1] || I.sure[x * 2] != 0.0 || I.sure[x * 2 + 1] != 0.0 || I.hw[x] != 0.5)
                blank = false;
        if (blank && share_blank) continue;
        T.row_of[r] = T.n_rows++;
        T.allele.insert(T.allele.end(), I.allele.begin(), I.allele.end());
        T.sure.insert(T.sure.end(), I.sure.begin(), I.sure.end());
        T.hw.insert(T.hw.end(), I.hw.begin(), I.hw.end());
    }
}

}  // namespace cnf2host
