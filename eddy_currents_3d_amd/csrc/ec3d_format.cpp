// ec3d_format.cpp — host side of the device matrix format: CSR (reference layout,
// src/EC3D.f90:36-38, 1-based) -> DIA bands + sliced-ELL tail, and back.
//
// Summation order is the contract (src/solvers.f90:59 sums a row in stored order): for each row the
// leading run of entries that lie on a band, with strictly ascending band index, goes to the bands;
// everything from the first entry that breaks that pattern goes to the row's tail in stored order.
// Bands are visited in ascending offset order and the tail after them, so the device row sum adds
// the same products in the same order (band slots a row does not use hold 0.0 and add +0).
#include "ec3d_internal.hpp"

#include <algorithm>
#include <cstring>
#include <unordered_map>

static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

int ec3d_csr_to_host_matrix(int64_t n, const double *valA, const int32_t *irow, const int32_t *jcol,
                            HostMatrix &M)
{
    if (n <= 0 || irow[0] != 1) {
        ec3d_set_error("ec3d_set_matrix_csr: need n > 0 and 1-based irow (irow[0] == 1)");
        return 2;
    }
    const int64_t nnz = (int64_t)irow[n] - 1;
    M = HostMatrix();
    M.n = n;
    M.nnz = nnz;
    M.n_pad = round_up(n, EC3D_TILE);

    // 1. band discovery on a row sample: offsets carried by >= 40 % of the rows
    {
        const int64_t stride = std::max<int64_t>(1, n / (1 << 20));
        std::unordered_map<int64_t, int64_t> cnt;
        int64_t rows = 0;
        for (int64_t r = 0; r < n; r += stride, ++rows)
            for (int64_t p = irow[r] - 1; p < irow[r + 1] - 1; ++p) ++cnt[(int64_t)jcol[p] - 1 - r];
        std::vector<std::pair<int64_t, int64_t>> cand; // (count, offset)
        for (auto &kv : cnt)
            if (kv.second * 10 >= rows * 4) cand.push_back({kv.second, kv.first});
        std::sort(cand.begin(), cand.end(), [](auto &a, auto &b) { return a.first > b.first; });
        if (cand.size() > EC3D_MAXB) cand.resize(EC3D_MAXB);
        M.nb = (int)cand.size();
        std::vector<int64_t> offs;
        for (auto &c : cand) offs.push_back(c.second);
        std::sort(offs.begin(), offs.end());
        for (int b = 0; b < M.nb; ++b) M.off[b] = offs[b];
    }
    auto band_of = [&](int64_t d) -> int {
        for (int b = 0; b < M.nb; ++b)
            if (M.off[b] == d) return b;
        return -1;
    };

    // 2. split rows
    M.bands.assign((size_t)M.nb * M.n_pad, 0.0);
    M.tail_id.assign((size_t)M.n_pad, -1);
    M.tile_flag.assign((size_t)(M.n_pad / EC3D_TILE), 0);
    std::vector<int64_t> tail_start; // per tail row: first CSR position of its tail
    std::vector<int64_t> tail_row;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t p0 = irow[r] - 1, p1 = irow[r + 1] - 1;
        int last_b = -1;
        int64_t p = p0;
        for (; p < p1; ++p) {
            const int64_t col = (int64_t)jcol[p] - 1;
            if (col < 0 || col >= n) {
                ec3d_set_error("ec3d_set_matrix_csr: column index out of range");
                return 2;
            }
            const int b = band_of(col - r);
            if (b < 0 || b <= last_b) break;
            M.bands[(size_t)b * M.n_pad + r] = valA[p];
            last_b = b;
        }
        if (p < p1) {
            for (int64_t q = p; q < p1; ++q) {
                const int64_t col = (int64_t)jcol[q] - 1;
                if (col < 0 || col >= n) {
                    ec3d_set_error("ec3d_set_matrix_csr: column index out of range");
                    return 2;
                }
            }
            M.tail_id[r] = (int32_t)tail_row.size();
            tail_row.push_back(r);
            tail_start.push_back(p);
            M.tile_flag[r / EC3D_TILE] = 1;
        }
    }

    // 3. sliced ELL over the tail rows (64 consecutive tail rows per slice, column-major inside)
    M.ntail = (int64_t)tail_row.size();
    const int64_t nchunk = (M.ntail + EC3D_CHUNK - 1) / EC3D_CHUNK;
    M.chunk_ptr.assign((size_t)nchunk + 1, 0);
    for (int64_t c = 0; c < nchunk; ++c) {
        int64_t w = 0;
        for (int64_t t = c * EC3D_CHUNK; t < std::min(M.ntail, (c + 1) * EC3D_CHUNK); ++t)
            w = std::max<int64_t>(w, (int64_t)irow[tail_row[t] + 1] - 1 - tail_start[t]);
        M.chunk_ptr[c + 1] = M.chunk_ptr[c] + w * EC3D_CHUNK;
    }
    M.tcol.assign((size_t)M.chunk_ptr[nchunk], 0);
    M.tval.assign((size_t)M.chunk_ptr[nchunk], 0.0);
    for (int64_t t = 0; t < M.ntail; ++t) {
        const int64_t base = M.chunk_ptr[t / EC3D_CHUNK] + (t % EC3D_CHUNK);
        const int64_t p1 = (int64_t)irow[tail_row[t] + 1] - 1;
        int64_t j = 0;
        for (int64_t p = tail_start[t]; p < p1; ++p, ++j) {
            M.tcol[(size_t)(base + j * EC3D_CHUNK)] = jcol[p] - 1;
            M.tval[(size_t)(base + j * EC3D_CHUNK)] = valA[p];
        }
    }
    return 0;
}

// Dictionary form of the bands: rows whose nb coefficients are bitwise equal share a class.
// Returns the number of classes, or 0 (and leaves M untouched) when more than 256 are needed.
int ec3d_build_dictionary_host(HostMatrix &M)
{
    if (M.nb != 7) return 0; // the specialised kernels cover the 7-band operator
    struct Key {
        uint64_t w[EC3D_MAXB];
        int nb;
        bool operator==(const Key &o) const { return memcmp(w, o.w, sizeof(uint64_t) * nb) == 0; }
    };
    struct Hash {
        size_t operator()(const Key &k) const
        {
            uint64_t h = 1469598103934665603ull;
            for (int b = 0; b < k.nb; ++b) h = (h ^ k.w[b]) * 1099511628211ull;
            return (size_t)h;
        }
    };
    std::unordered_map<Key, int, Hash> dict;
    std::vector<uint8_t> cls((size_t)M.n_pad, 0);
    std::vector<double> table;
    Key last{};
    int last_id = -1;
    for (int64_t r = 0; r < M.n_pad; ++r) {
        Key k{};
        k.nb = M.nb;
        for (int b = 0; b < M.nb; ++b) memcpy(&k.w[b], &M.bands[(size_t)b * M.n_pad + r], 8);
        int id;
        if (last_id >= 0 && k == last) {
            id = last_id;
        } else {
            auto it = dict.find(k);
            if (it == dict.end()) {
                if (dict.size() == 256) return 0;
                id = (int)dict.size();
                dict.emplace(k, id);
                for (int b = 0; b < M.nb; ++b) table.push_back(M.bands[(size_t)b * M.n_pad + r]);
            } else {
                id = it->second;
            }
            last = k;
            last_id = id;
        }
        cls[(size_t)r] = (uint8_t)id;
    }
    M.cls.swap(cls);
    M.table.swap(table);
    M.ncls = (int)dict.size();
    return M.ncls;
}

// inverse, for parity checks of the device assembly.  Band slots holding exactly 0.0 and tail
// padding (value 0.0) are not emitted, so explicit zeros of a source CSR do not round-trip.
void ec3d_host_matrix_to_csr(const HostMatrix &M, std::vector<int32_t> &irow, std::vector<int32_t> &jcol,
                             std::vector<double> &valA)
{
    irow.assign((size_t)M.n + 1, 0);
    jcol.clear();
    valA.clear();
    irow[0] = 1;
    for (int64_t r = 0; r < M.n; ++r) {
        for (int b = 0; b < M.nb; ++b) {
            const double v = M.bands[(size_t)b * M.n_pad + r];
            if (v != 0.0) {
                jcol.push_back((int32_t)(r + M.off[b] + 1));
                valA.push_back(v);
            }
        }
        const int32_t t = M.tail_id[r];
        if (t >= 0) {
            const int64_t base = M.chunk_ptr[t / EC3D_CHUNK], end = M.chunk_ptr[t / EC3D_CHUNK + 1];
            for (int64_t e = base + t % EC3D_CHUNK; e < end; e += EC3D_CHUNK)
                if (M.tval[(size_t)e] != 0.0) {
                    jcol.push_back(M.tcol[(size_t)e] + 1);
                    valA.push_back(M.tval[(size_t)e]);
                }
        }
        irow[r + 1] = (int32_t)(jcol.size() + 1);
    }
}

bool ec3d_host_matrix_is_cube(const HostMatrix &M, int64_t &sdx, int64_t &kdz)
{
    if (M.ntail != 0 || M.nb != 7 || M.off[3] != 0 || M.off[2] != -1 || M.off[4] != 1 || M.off[1] != -M.off[5] ||
        M.off[0] != -M.off[6] || M.off[5] < 2 || M.off[6] <= M.off[5] || M.off[6] % M.off[5] != 0 ||
        M.n % M.off[6] != 0)
        return false;
    sdx = M.off[5];
    kdz = M.off[6];
    return true;
}
