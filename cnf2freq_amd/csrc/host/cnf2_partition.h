// cnf2_partition.h -- who sweeps and who updates what in a multi-process haplotyping run (the
// reference's dead MPI code deals individuals to ranks at cnF2freq.cpp:5297-5299 and reduces per individual at 6245-6254).
// A pure function of the window table (cnf2_window_table of cnf2hip.h: per analysed individual the records in its 7 window
// slots and their tie groups), so every rank computes the same plan, and CPU tests can check it (tests/shim).
#ifndef CNF2_PARTITION_H
#define CNF2_PARTITION_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace cnf2host {

// Blocks of analysed individuals are contiguous in dous and balanced by cost, their boundaries moved (within a tolerance)
// to where the fewest records are touched from both sides, so families that fit inside a block stay whole and nothing of
// them is ever exchanged.  A record touched by the windows of one rank only is PRIVATE to it; a record touched by several
// is SHARED and owned by one of them.  A rank updates the records it owns (cnF2freq.cpp:6344-6368 loops over individuals;
// an update reads only the individual's own accumulators and rows).  Records no window touches collect no evidence: their
// update is a no-op (no infprobs, haplocount 0) and nobody owns them.
struct Partition {
    int rank = 0, world = 1;
    std::vector<int>                  bounds;      // [world + 1] boundaries of the blocks in dous
    std::vector<std::vector<int32_t>> shared_of;   // per owner: the shared records it owns, ascending
    std::vector<std::vector<int32_t>> private_of;  // per rank: the records only its windows touch, ascending
    std::vector<int32_t>              owned;       // private_of[rank] + shared_of[rank], ascending
    size_t seg_shared = 0, seg_private = 0;        // records per segment of the exchange buffers (max over ranks)
    size_t n_shared = 0;                           // shared records in all
};

// w = window table [N][17]: [3..9] slot records (-1 none), [10..16] tie group per slot (-1 none); R records, M markers
inline Partition plan_partition(int R, int N, int M, const int32_t* w, int rank, int world)
{
    Partition Q;
    Q.rank = rank;
    Q.world = world;
    std::vector<double> cost(N), prefix(N + 1, 0.0);
    for (int j = 0; j < N; j++) {
        int groups = 0;
        for (int k = 0; k < 7; k++) groups = std::max(groups, w[(size_t)j * 17 + 10 + k] + 1);
        cost[j] = (double)M * (1.0 + (double)(1 << groups));
        prefix[j + 1] = prefix[j] + cost[j];
    }
    const double total = prefix[N];
    // straddle[j] = records touched by windows on both sides of a cut between dous[j - 1] and dous[j]
    std::vector<int> first(R, N), last(R, -1);
    for (int j = 0; j < N; j++)
        for (int k = 0; k < 7; k++) {
            const int r = w[(size_t)j * 17 + 3 + k];
            if (r < 0) continue;
            first[r] = std::min(first[r], j);
            last[r]  = std::max(last[r], j);
        }
    std::vector<int> straddle(N + 2, 0);
    for (int r = 0; r < R; r++)
        if (last[r] > first[r]) {
            straddle[first[r] + 1]++;
            straddle[last[r] + 1]--;
        }
    for (int j = 1; j <= N; j++) straddle[j] += straddle[j - 1];
    // boundary k: the cost-balanced position (first individual whose prefix cost reaches k / world of the total), moved
    // to the position within a quarter of a block's cost of it that the fewest records straddle (the nearest such position:
    // with families much smaller than a block the shift is half a family)
    Q.bounds.assign(world + 1, 0);
    Q.bounds[world] = N;
    const double tol = 0.25 * total / world;
    for (int k = 1; k < world; k++) {
        const double target = total * k / world;
        int ideal = N;
        for (int j = 0; j < N; j++)
            if (prefix[j] + 0.5 * cost[j] >= target) {
                ideal = j;
                break;
            }
        int best = ideal;
        for (int j = std::max(Q.bounds[k - 1], 0); j <= N; j++) {
            if (fabs(prefix[j] - prefix[ideal]) > tol) continue;
            const int sj = (j == 0 || j == N) ? 0 : straddle[j], sb = (best == 0 || best == N) ? 0 : straddle[best];
            if (sj < sb || (sj == sb && abs(j - ideal) < abs(best - ideal))) best = j;
        }
        Q.bounds[k] = std::max(best, Q.bounds[k - 1]);
    }
    // who touches what
    std::vector<int>     touchers(R, 0), one(R, -1);
    std::vector<uint8_t> seen((size_t)R, 0);
    std::vector<std::vector<int32_t>> touched(world);
    for (int q = 0; q < world; q++) {
        for (int j = Q.bounds[q]; j < Q.bounds[q + 1]; j++)
            for (int k = 0; k < 7; k++) {
                const int r = w[(size_t)j * 17 + 3 + k];
                if (r < 0 || seen[r]) continue;
                seen[r] = 1;
                touched[q].push_back(r);
            }
        for (int r : touched[q]) {
            seen[r] = 0;
            touchers[r]++;
            one[r] = q;
        }
        std::sort(touched[q].begin(), touched[q].end());
    }
    Q.shared_of.assign(world, std::vector<int32_t>());
    Q.private_of.assign(world, std::vector<int32_t>());
    // a shared record goes to the rank among its touchers that owns the fewest so far (ties: the lowest rank); records in
    // ascending order, so every rank computes the same owners
    std::vector<std::vector<int>> who(R);
    for (int q = 0; q < world; q++)
        for (int r : touched[q])
            if (touchers[r] > 1) who[r].push_back(q);
    for (int r = 0; r < R; r++) {
        if (touchers[r] == 1) Q.private_of[one[r]].push_back(r);
        else if (touchers[r] > 1) {
            int own = who[r][0];
            for (int q : who[r])
                if (Q.shared_of[q].size() < Q.shared_of[own].size()) own = q;
            Q.shared_of[own].push_back(r);
            Q.n_shared++;
        }
    }
    for (int q = 0; q < world; q++) {
        Q.seg_shared = std::max(Q.seg_shared, Q.shared_of[q].size());
        Q.seg_private = std::max(Q.seg_private, Q.private_of[q].size());
    }
    Q.owned = Q.private_of[rank];
    Q.owned.insert(Q.owned.end(), Q.shared_of[rank].begin(), Q.shared_of[rank].end());
    std::sort(Q.owned.begin(), Q.owned.end());
    return Q;
}


}  // namespace cnf2host
#endif
