// cnf2_text.h -- the two texts a run writes per individual, formed in memory (cnf2_format.h) so that the host's threads can
// form them side by side: the per-locus rows of an analysed individual on one chromosome (cnF2freq.cpp:6183-6188) and an
// individual's part of the haplotype dump (cnF2freq.cpp:8157-8192).  Character for character what the reference's fprintf
// calls write (tests/test_host_format.py renders both ways and compares).
#ifndef CNF2_TEXT_H
#define CNF2_TEXT_H

#include "cnf2_format.h"
#include "cnf2_readers.h"

namespace cnf2host {

// "%s:%d\n", then -- unless the individual was skipped on this chromosome (cnF2freq.cpp:5403) -- "%.5lf\t%.5lf\t%.5lf\n" for
// the markers [m0, m1) (`dosage` = the individual's [M][3] rows), then an empty line
inline void rows_text(const std::string& name, int chrom1, const double* dosage, int m0, int m1, bool skipped, TextBuf& tb)
{
    tb.put(name.data(), name.size());
    tb.put(':');
    tb.integer(chrom1);
    tb.put('\n');
    if (!skipped)
        for (int m = m0; m < m1; m++) {
            const double* d = dosage + (size_t)m * 3;
            tb.fixed(d[0], 5);
            tb.put('\t');
            tb.fixed(d[1], 5);
            tb.put('\t');
            tb.fixed(d[2], 5);
            tb.put('\n');
        }
    tb.put('\n');
}

// "%d %s\n" and per marker "%f\t%d\t%d\t\t%f\t%lf %lf %lf": haploweight, the two alleles, 0.0 (a constant in the reference
// too), the two certainties, 0.5; then "\t%d\t%d\t%lf\t%lf" (the priors) where the individual has any; "\n"
inline void dump_text(const Individual& I, int M, TextBuf& tb)
{
    tb.integer(I.n);
    tb.put(' ');
    tb.put(I.name.data(), I.name.size());
    tb.put('\n');
    for (int m = 0; m < M; m++) {
        tb.fixed(I.hw[m], 6);
        tb.put('\t');
        tb.integer(I.allele[m * 2]);
        tb.put('\t');
        tb.integer(I.allele[m * 2 + 1]);
        tb.put("\t\t0.000000\t", 11);
        tb.fixed(I.sure[m * 2], 6);
        tb.put(' ');
        tb.fixed(I.sure[m * 2 + 1], 6);
        tb.put(" 0.500000", 9);
        if (I.has_prior) {
            tb.put('\t');
            tb.integer(I.prior_allele[m * 2]);
            tb.put('\t');
            tb.integer(I.prior_allele[m * 2 + 1]);
            tb.put('\t');
            tb.fixed(I.prior_sure[m * 2], 6);
            tb.put('\t');
            tb.fixed(I.prior_sure[m * 2 + 1], 6);
        }
        tb.put('\n');
    }
}

}  // namespace cnf2host
#endif
