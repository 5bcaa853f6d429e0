// cnf2_readers.h -- host-side data model and PlantImpute-format readers of the drop-in
// `cnF2freq` command line (plain C++17, no Boost).
//
// Mirrors, with their quirks: getind / individ creation (cnF2freq.cpp:2448-2514, 6479-6493),
// readalphamap (6669-6685), readalphaped (6495-6540), readalphadata (6542-6667).
#ifndef CNF2_READERS_H
#define CNF2_READERS_H

#include <stdint.h>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

namespace cnf2host {

// One `individ` (cnF2freq.cpp:853-914) restricted to what the sweep, the dump and the readers use.
struct Individual {
    int         n = 0;             // the reference's 1-based number (order of first mention)
    std::string name;
    int         gen = 0;
    bool        empty = false;
    int         pars[2] = {-1, -1};  // index into Pedigree::inds, -1 = none
    bool        has_prior = false;   // priormarkerdata set (a genotype line was read)
    std::vector<uint8_t> allele;     // [M][2] MarkerVal: 0 unknown, 1, 2, 9
    std::vector<double>  sure;       // [M][2]
    std::vector<double>  hw;         // [M]
    std::vector<uint8_t> prior_allele;
    std::vector<double>  prior_sure;
};

struct Pedigree {
    std::vector<double>         pos;          // markerposes
    std::vector<int32_t>        chromstarts;  // incl. the closing entry
    std::vector<Individual>     inds;         // inds[i].n == i + 1
    std::map<std::string, int>  index;        // name -> index, "0" -> -1 (cnF2freq.cpp:6493)
    std::vector<int>            dous;         // analysed individuals (gen >= 2), file order
    std::vector<std::string>    log;          // what the reference prints to stdout while reading

    int  getind(const std::string& name);     // creates on first mention; returns -1 for "0"
    int  n_markers() const { return (int)pos.size(); }
};

bool read_alpha_map(FILE* in, Pedigree& P);
bool read_alpha_ped(FILE* in, Pedigree& P);
bool read_alpha_gen(FILE* in, Pedigree& P);
// --capmarker n: keep the first n markers (intent of cnF2freq.cpp:7965-7969)
void cap_markers(Pedigree& P, int cap);

// Flattened tables for cnf2_upload_rows / cnf2_upload_pedigree.  Row 0 is the shared blank row;
// every individual whose data equals the blank row maps to it.
struct Tables {
    std::vector<int32_t> par, gen, row_of, dous;
    std::vector<uint8_t> empty;
    std::vector<uint8_t> allele;   // [rows][M][2]
    std::vector<double>  sure;     // [rows][M][2]
    std::vector<double>  hw;       // [rows][M]
    int                  n_rows = 0;
};
// share_blank: every individual whose data equals the blank row maps to row 0 (a single sweep); otherwise one row per
// individual, row_of[r] = r + 1 (runs that write rows in place)
void build_tables(const Pedigree& P, Tables& T, bool share_blank = true);

}  // namespace cnf2host
#endif
