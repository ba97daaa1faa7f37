// pcd_io.hpp — self-contained point-cloud I/O for the host (Open3D is not available here): PCD (ascii / binary /
// binary_compressed) and PLY (ascii / binary_little_endian) readers that return AoS fp64 xyz with NaN / inf points
// removed (the reference reads with remove_nan = remove_infinite = true, map_eval.cpp:6), and a binary PCD writer.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pcio {

struct Field {
    std::string name;
    int size = 4;
    char type = 'F';
    int count = 1;
    size_t offset = 0;  // byte offset inside one point record
};

inline double read_scalar(const unsigned char *p, char type, int size) {
    switch (type) {
        case 'F':
            if (size == 4) { float v; std::memcpy(&v, p, 4); return v; }
            if (size == 8) { double v; std::memcpy(&v, p, 8); return v; }
            break;
        case 'I':
            if (size == 1) { int8_t v; std::memcpy(&v, p, 1); return v; }
            if (size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
            if (size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; }
            if (size == 8) { int64_t v; std::memcpy(&v, p, 8); return (double) v; }
            break;
        case 'U':
            if (size == 1) { uint8_t v; std::memcpy(&v, p, 1); return v; }
            if (size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
            if (size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; }
            if (size == 8) { uint64_t v; std::memcpy(&v, p, 8); return (double) v; }
            break;
    }
    throw std::runtime_error("unsupported PCD field type/size");
}

// liblzf-format decompressor (the format PCL's binary_compressed PCD uses)
inline size_t lzf_decompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
    const unsigned char *ip = in, *const in_end = in + in_len;
    unsigned char *op = out, *const out_end = out + out_len;
    while (ip < in_end) {
        unsigned int ctrl = *ip++;
        if (ctrl < (1u << 5)) {  // literal run
            ctrl++;
            if (op + ctrl > out_end || ip + ctrl > in_end) return 0;
            std::memcpy(op, ip, ctrl);
            op += ctrl;
            ip += ctrl;
        } else {  // back reference
            unsigned int len = ctrl >> 5;
            if (ip >= in_end) return 0;
            if (len == 7) {
                len += *ip++;
                if (ip >= in_end) return 0;
            }
            const unsigned char *ref = op - ((ctrl & 0x1f) << 8) - 1 - *ip++;
            if (ref < out || op + len + 2 > out_end) return 0;
            len += 2;
            for (unsigned int k = 0; k < len; ++k) *op++ = *ref++;  // may overlap: byte by byte
        }
    }
    return (size_t) (op - out);
}

inline bool push_if_finite(std::vector<double> &xyz, double x, double y, double z) {
    if (std::isfinite(x) && std::isfinite(y) && std::isfinite(z)) {
        xyz.push_back(x);
        xyz.push_back(y);
        xyz.push_back(z);
        return true;
    }
    return false;
}

// normals (optional): filled with normal_x / normal_y / normal_z of the kept points when the file has those fields
// (Open3D's reader does the same: PointCloud::HasNormals() is what point-to-plane ICP asks for), left empty otherwise.
inline bool read_pcd(const std::string &path, std::vector<double> &xyz, std::string *err = nullptr,
                     std::vector<double> *normals = nullptr) {
    auto fail = [&](const std::string &m) {
        if (err) *err = m;
        return false;
    };
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return fail("cannot open " + path);
    std::vector<Field> fields;
    std::vector<std::string> names;
    std::vector<int> sizes, counts;
    std::vector<char> types;
    size_t n_points = 0, width = 0, height = 1;
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::stringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "FIELDS" || key == "COLUMNS") { std::string s; while (ss >> s) names.push_back(s); }
        else if (key == "SIZE") { int v; while (ss >> v) sizes.push_back(v); }
        else if (key == "TYPE") { char c; while (ss >> c) types.push_back(c); }
        else if (key == "COUNT") { int v; while (ss >> v) counts.push_back(v); }
        else if (key == "WIDTH") ss >> width;
        else if (key == "HEIGHT") ss >> height;
        else if (key == "POINTS") ss >> n_points;
        else if (key == "DATA") { ss >> data_kind; break; }
    }
    if (names.empty() || data_kind.empty()) return fail("not a PCD file: " + path);
    if (n_points == 0) n_points = width * height;
    if (counts.empty()) counts.assign(names.size(), 1);
    if (sizes.size() != names.size() || types.size() != names.size() || counts.size() != names.size())
        return fail("inconsistent PCD header in " + path);
    size_t rec = 0;
    int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
    for (size_t i = 0; i < names.size(); ++i) {
        Field fd;
        fd.name = names[i];
        fd.size = sizes[i];
        fd.type = types[i];
        fd.count = counts[i];
        fd.offset = rec;
        rec += (size_t) fd.size * fd.count;
        if (fd.name == "x") ix = (int) i;
        if (fd.name == "y") iy = (int) i;
        if (fd.name == "z") iz = (int) i;
        if (fd.name == "normal_x") inx = (int) i;
        if (fd.name == "normal_y") iny = (int) i;
        if (fd.name == "normal_z") inz = (int) i;
        fields.push_back(fd);
    }
    if (ix < 0 || iy < 0 || iz < 0) return fail("PCD has no x/y/z fields: " + path);
    xyz.clear();
    xyz.reserve(n_points * 3);
    const bool want_n = normals && inx >= 0 && iny >= 0 && inz >= 0;
    if (normals) normals->clear();
    if (want_n) normals->reserve(n_points * 3);
    auto push_n = [&](double a, double b, double c) {
        normals->push_back(a);
        normals->push_back(b);
        normals->push_back(c);
    };
    if (data_kind == "ascii") {
        size_t total_cols = 0;
        std::vector<size_t> col0(fields.size());
        for (size_t i = 0; i < fields.size(); ++i) { col0[i] = total_cols; total_cols += fields[i].count; }
        std::vector<double> row(total_cols);
        for (size_t p = 0; p < n_points && std::getline(f, line); ++p) {
            std::stringstream ss(line);
            bool ok = true;
            for (size_t c = 0; c < total_cols; ++c) {
                std::string tok;
                if (!(ss >> tok)) { ok = false; break; }
                row[c] = (tok == "nan" || tok == "NaN") ? NAN : std::strtod(tok.c_str(), nullptr);
            }
            if (ok && push_if_finite(xyz, row[col0[ix]], row[col0[iy]], row[col0[iz]]) && want_n)
                push_n(row[col0[inx]], row[col0[iny]], row[col0[inz]]);
        }
    } else if (data_kind == "binary") {
        std::vector<unsigned char> buf(rec * n_points);
        f.read(reinterpret_cast<char *>(buf.data()), (std::streamsize) buf.size());
        if ((size_t) f.gcount() != buf.size()) return fail("truncated binary PCD: " + path);
        for (size_t p = 0; p < n_points; ++p) {
            const unsigned char *r = buf.data() + p * rec;
            auto at = [&](int fi) { return read_scalar(r + fields[fi].offset, fields[fi].type, fields[fi].size); };
            if (push_if_finite(xyz, at(ix), at(iy), at(iz)) && want_n) push_n(at(inx), at(iny), at(inz));
        }
    } else if (data_kind == "binary_compressed") {
        uint32_t csize = 0, usize = 0;
        f.read(reinterpret_cast<char *>(&csize), 4);
        f.read(reinterpret_cast<char *>(&usize), 4);
        std::vector<unsigned char> cbuf(csize), ubuf(usize);
        f.read(reinterpret_cast<char *>(cbuf.data()), csize);
        if ((size_t) f.gcount() != csize) return fail("truncated compressed PCD: " + path);
        if (lzf_decompress(cbuf.data(), csize, ubuf.data(), usize) != usize) return fail("LZF decompression failed: " + path);
        // layout after decompression: field by field (structure of arrays)
        std::vector<size_t> base(fields.size());
        size_t off = 0;
        for (size_t i = 0; i < fields.size(); ++i) { base[i] = off; off += (size_t) fields[i].size * fields[i].count * n_points; }
        if (off > usize) return fail("compressed PCD payload too small: " + path);
        auto at = [&](int fi, size_t p) {
            return read_scalar(ubuf.data() + base[fi] + p * (size_t) fields[fi].size * fields[fi].count, fields[fi].type, fields[fi].size);
        };
        for (size_t p = 0; p < n_points; ++p)
            if (push_if_finite(xyz, at(ix, p), at(iy, p), at(iz, p)) && want_n) push_n(at(inx, p), at(iny, p), at(inz, p));
    } else {
        return fail("unsupported PCD DATA kind '" + data_kind + "'");
    }
    return true;
}

inline int ply_type_size(const std::string &t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64" || t == "int64" || t == "uint64") return 8;
    return -1;
}

// normals (optional): the vertex properties nx / ny / nz of the kept points, when the file has them
inline bool read_ply(const std::string &path, std::vector<double> &xyz, std::string *err = nullptr,
                     std::vector<double> *normals = nullptr) {
    auto fail = [&](const std::string &m) {
        if (err) *err = m;
        return false;
    };
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return fail("cannot open " + path);
    std::string line, format;
    std::getline(f, line);
    if (line.substr(0, 3) != "ply") return fail("not a PLY file: " + path);
    size_t n_vertex = 0;
    bool in_vertex = false, vertex_first = true, seen_element = false;
    struct Prop { std::string type, name; };
    std::vector<Prop> props;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::stringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "format") ss >> format;
        else if (key == "element") {
            std::string name;
            size_t cnt;
            ss >> name >> cnt;
            in_vertex = (name == "vertex");
            if (in_vertex) { n_vertex = cnt; vertex_first = !seen_element; }
            seen_element = true;
        } else if (key == "property" && in_vertex) {
            Prop p;
            ss >> p.type;
            if (p.type == "list") return fail("list property in vertex element not supported");
            ss >> p.name;
            props.push_back(p);
        } else if (key == "end_header") break;
    }
    if (!vertex_first) return fail("PLY: vertex element must come first");
    int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
    std::vector<size_t> off(props.size());
    size_t rec = 0;
    for (size_t i = 0; i < props.size(); ++i) {
        const int sz = ply_type_size(props[i].type);
        if (sz < 0) return fail("PLY: unknown property type " + props[i].type);
        off[i] = rec;
        rec += (size_t) sz;
        if (props[i].name == "x") ix = (int) i;
        if (props[i].name == "y") iy = (int) i;
        if (props[i].name == "z") iz = (int) i;
        if (props[i].name == "nx") inx = (int) i;
        if (props[i].name == "ny") iny = (int) i;
        if (props[i].name == "nz") inz = (int) i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return fail("PLY has no x/y/z vertex properties");
    xyz.clear();
    xyz.reserve(n_vertex * 3);
    const bool want_n = normals && inx >= 0 && iny >= 0 && inz >= 0;
    if (normals) normals->clear();
    if (want_n) normals->reserve(n_vertex * 3);
    auto push_n = [&](double a, double b, double c) {
        normals->push_back(a);
        normals->push_back(b);
        normals->push_back(c);
    };
    if (format == "ascii") {
        std::vector<double> row(props.size());
        for (size_t p = 0; p < n_vertex && std::getline(f, line); ++p) {
            std::stringstream ss(line);
            std::string tok;
            for (auto &v : row) {  // strtod: "nan" / "inf" tokens parse as such (operator>> would fail on them)
                v = NAN;
                if (ss >> tok) v = std::strtod(tok.c_str(), nullptr);
            }
            if (push_if_finite(xyz, row[ix], row[iy], row[iz]) && want_n) push_n(row[inx], row[iny], row[inz]);
        }
    } else if (format == "binary_little_endian") {
        std::vector<unsigned char> buf(rec * n_vertex);
        f.read(reinterpret_cast<char *>(buf.data()), (std::streamsize) buf.size());
        if ((size_t) f.gcount() != buf.size()) return fail("truncated binary PLY: " + path);
        auto get = [&](const unsigned char *r, int i) -> double {
            const std::string &t = props[i].type;
            const int sz = ply_type_size(t);
            const char kind = (t[0] == 'f' || t[0] == 'd') ? 'F' : ((t[0] == 'u') ? 'U' : 'I');
            return read_scalar(r + off[i], kind, sz);
        };
        for (size_t p = 0; p < n_vertex; ++p) {
            const unsigned char *r = buf.data() + p * rec;
            if (push_if_finite(xyz, get(r, ix), get(r, iy), get(r, iz)) && want_n) push_n(get(r, inx), get(r, iny), get(r, inz));
        }
    } else {
        return fail("unsupported PLY format '" + format + "'");
    }
    return true;
}

// binary PCD writer: x y z as float64 (8-byte F fields) and an optional packed-float rgb column
inline bool write_pcd(const std::string &path, const double *xyz, size_t n, const float *rgb_packed = nullptr) {
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n";
    if (rgb_packed) f << "FIELDS x y z rgb\nSIZE 8 8 8 4\nTYPE F F F F\nCOUNT 1 1 1 1\n";
    else f << "FIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\n";
    f << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    for (size_t i = 0; i < n; ++i) {
        f.write(reinterpret_cast<const char *>(xyz + 3 * i), 24);
        if (rgb_packed) f.write(reinterpret_cast<const char *>(rgb_packed + i), 4);
    }
    return (bool) f;
}

}  // namespace pcio
