// yaml_lite.hpp — the YAML subset MapEval's config files use (map_eval/config/*.yaml), no third-party dependency:
//   key: scalar            # trailing comments allowed
//   key: [a, b, c]         # flow sequence of scalars
//   key:                   # block sequence of flow sequences (initial_matrix)
//     - [a, b, c, d]
// yaml-cpp is not available here; the reference's loader (map_eval_main.cpp:120-208) only ever reads these forms.
#pragma once

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace yaml_lite {

struct Node {
    bool defined = false;
    std::string scalar;                            // key: scalar
    std::vector<std::string> seq;                  // key: [a, b]
    std::vector<std::vector<std::string>> rows;    // key:\n  - [..]\n  - [..]
};

inline std::string trim(const std::string &s) {
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    size_t e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
}

inline std::string strip_comment(const std::string &line) {
    bool in_s = false, in_d = false;
    for (size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (c == '\'' && !in_d) in_s = !in_s;
        else if (c == '"' && !in_s) in_d = !in_d;
        else if (c == '#' && !in_s && !in_d && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) return line.substr(0, i);
    }
    return line;
}

inline std::string unquote(const std::string &s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
        return s.substr(1, s.size() - 2);
    return s;
}

inline std::vector<std::string> parse_flow(const std::string &s) {
    std::vector<std::string> out;
    const size_t b = s.find('['), e = s.rfind(']');
    if (b == std::string::npos || e == std::string::npos || e < b) throw std::runtime_error("malformed flow sequence: " + s);
    std::stringstream ss(s.substr(b + 1, e - b - 1));
    std::string item;
    while (std::getline(ss, item, ',')) {
        item = trim(item);
        if (!item.empty()) out.push_back(unquote(item));
    }
    return out;
}

class Document {
public:
    static Document load_file(const std::string &path) {
        std::ifstream f(path);
        if (!f.is_open()) throw std::runtime_error("bad file: " + path);  // YAML::BadFile
        Document d;
        std::string line, cur_key;
        while (std::getline(f, line)) {
            std::string body = strip_comment(line);
            if (trim(body).empty()) continue;
            const size_t indent = body.find_first_not_of(" \t");
            const std::string t = trim(body);
            if (t[0] == '-' && indent > 0 && !cur_key.empty()) {  // block-sequence item of the current key
                d.nodes_[cur_key].rows.push_back(parse_flow(t.substr(1)));
                continue;
            }
            const size_t colon = t.find(':');
            if (colon == std::string::npos) throw std::runtime_error("cannot parse line: " + line);
            const std::string key = trim(t.substr(0, colon));
            const std::string val = trim(t.substr(colon + 1));
            Node &n = d.nodes_[key];
            n = Node();
            n.defined = true;
            cur_key = key;
            if (val.empty()) continue;  // block sequence follows
            if (val[0] == '[') n.seq = parse_flow(val);
            else n.scalar = unquote(val);
        }
        return d;
    }

    bool has(const std::string &k) const { return nodes_.count(k) != 0; }
    const Node &at(const std::string &k) const {
        auto it = nodes_.find(k);
        if (it == nodes_.end()) throw std::runtime_error("missing required key: " + k);  // yaml-cpp: bad conversion on a null node
        return it->second;
    }
    double as_double(const std::string &k) const { return to_double(at(k).scalar, k); }
    int as_int(const std::string &k) const { return (int) to_double(at(k).scalar, k); }
    std::string as_string(const std::string &k) const { return at(k).scalar; }
    bool as_bool(const std::string &k) const {
        const std::string v = at(k).scalar;
        if (v == "true" || v == "True" || v == "TRUE" || v == "yes" || v == "on") return true;
        if (v == "false" || v == "False" || v == "FALSE" || v == "no" || v == "off") return false;
        throw std::runtime_error("bad conversion to bool for key " + k + ": '" + v + "'");
    }
    static double to_double(const std::string &s, const std::string &k) {
        char *end = nullptr;
        const double v = std::strtod(s.c_str(), &end);
        if (s.empty() || end == s.c_str() || *end != '\0') throw std::runtime_error("bad conversion to number for key " + k + ": '" + s + "'");
        return v;
    }

private:
    std::map<std::string, Node> nodes_;
};

}  // namespace yaml_lite
