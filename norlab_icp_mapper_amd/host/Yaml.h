// Yaml.h -- reader for the YAML subset norlab_icp_mapper configuration files use
// (examples/config.yaml, docs/MapperConfiguration.md): block maps, block sequences of single-key
// maps, scalars, inline `[a, b]` lists, inline `{a: 1, b: 2}` maps, `#` comments.  yaml-cpp is not
// available in the build image; the reference parses the same files with it (Mapper.cpp:59-185).
#pragma once
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace nim {
namespace yaml {

struct Exception : std::runtime_error { using std::runtime_error::runtime_error; }; // YAML::Exception

class Node {
public:
    enum Kind { Null, Scalar, Map, Seq };
    Kind kind = Null;
    std::string scalar;
    std::vector<std::pair<std::string, Node>> map; // insertion order kept (module order matters)
    std::vector<Node> seq;

    bool IsNull() const { return kind == Null; }
    bool IsMap() const { return kind == Map; }
    bool IsSequence() const { return kind == Seq; }
    bool IsScalar() const { return kind == Scalar; }
    explicit operator bool() const { return kind != Null; }

    const Node& operator[](const std::string& key) const {
        static const Node none;
        if (kind != Map) return none;
        for (const auto& kv : map) if (kv.first == key) return kv.second;
        return none;
    }
    template <typename T> T as() const;
    std::string str() const { if (kind != Scalar) throw Exception("expected a scalar"); return scalar; }
};

template <> inline std::string Node::as<std::string>() const { return str(); }
template <> inline float Node::as<float>() const {
    const std::string s = str();
    if (s == "inf" || s == ".inf" || s == "+inf") return HUGE_VALF;
    if (s == "-inf" || s == "-.inf") return -HUGE_VALF;
    char* end = nullptr;
    const float v = std::strtof(s.c_str(), &end);
    if (end == s.c_str() || *end) throw Exception("bad float: " + s);
    return v;
}
template <> inline double Node::as<double>() const { return (double)as<float>(); }
template <> inline int Node::as<int>() const {
    const std::string s = str();
    char* end = nullptr;
    const long v = std::strtol(s.c_str(), &end, 10);
    if (end == s.c_str() || *end) throw Exception("bad integer: " + s);
    return (int)v;
}
template <> inline bool Node::as<bool>() const {
    const std::string s = str();
    if (s == "1" || s == "true" || s == "True") return true;
    if (s == "0" || s == "false" || s == "False") return false;
    throw Exception("bad boolean: " + s);
}

namespace detail {

struct Line { int indent; std::string text; };

inline std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
}

inline std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}

inline std::vector<std::string> split_top(const std::string& s, char sep) {
    std::vector<std::string> out;
    int depth = 0;
    std::string cur;
    for (char ch : s) {
        if (ch == '[' || ch == '{') ++depth;
        if (ch == ']' || ch == '}') --depth;
        if (ch == sep && depth == 0) { out.push_back(trim(cur)); cur.clear(); }
        else cur.push_back(ch);
    }
    if (!trim(cur).empty() || !out.empty()) out.push_back(trim(cur));
    return out;
}

inline size_t find_colon(const std::string& s) {
    int depth = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '[' || s[i] == '{') ++depth;
        if (s[i] == ']' || s[i] == '}') --depth;
        if (s[i] == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
    }
    return std::string::npos;
}

inline Node parse_inline(const std::string& text) {
    const std::string s = trim(text);
    Node n;
    if (s.empty() || s == "~" || s == "null") return n;
    if (s.front() == '[') {
        if (s.back() != ']') throw Exception("unterminated inline list: " + s);
        n.kind = Node::Seq;
        for (const auto& item : split_top(s.substr(1, s.size() - 2), ',')) if (!item.empty()) n.seq.push_back(parse_inline(item));
        return n;
    }
    if (s.front() == '{') {
        if (s.back() != '}') throw Exception("unterminated inline map: " + s);
        n.kind = Node::Map;
        for (const auto& item : split_top(s.substr(1, s.size() - 2), ',')) {
            if (item.empty()) continue;
            const size_t c = find_colon(item);
            if (c == std::string::npos) throw Exception("bad inline map entry: " + item);
            n.map.emplace_back(unquote(trim(item.substr(0, c))), parse_inline(item.substr(c + 1)));
        }
        return n;
    }
    n.kind = Node::Scalar;
    n.scalar = unquote(s);
    return n;
}

inline Node parse_block(std::vector<Line>& lines, size_t& i, int indent);

inline Node parse_value_after_key(std::vector<Line>& lines, size_t& i, int key_indent, const std::string& rest) {
    if (!trim(rest).empty()) return parse_inline(rest);
    if (i < lines.size() && lines[i].indent > key_indent) return parse_block(lines, i, lines[i].indent);
    // "key:" followed by a sequence at the same indent is also legal YAML
    if (i < lines.size() && lines[i].indent == key_indent && lines[i].text.rfind("- ", 0) == 0) return parse_block(lines, i, key_indent);
    return Node();
}

inline Node parse_block(std::vector<Line>& lines, size_t& i, int indent) {
    Node n;
    if (i >= lines.size()) return n;
    const bool is_seq = lines[i].text.rfind("- ", 0) == 0 || lines[i].text == "-";
    n.kind = is_seq ? Node::Seq : Node::Map;
    while (i < lines.size() && lines[i].indent == indent) {
        Line& ln = lines[i];
        if (is_seq) {
            if (!(ln.text.rfind("- ", 0) == 0 || ln.text == "-")) break;
            // turn "- item" into a nested block that starts at indent + 2
            const std::string item = ln.text.size() > 2 ? ln.text.substr(2) : "";
            const int inner = indent + 2 + (int)(item.size() - trim(item).size());
            if (trim(item).empty()) { ++i; n.seq.push_back(i < lines.size() && lines[i].indent > indent ? parse_block(lines, i, lines[i].indent) : Node()); continue; }
            if (find_colon(trim(item)) == std::string::npos) { ++i; n.seq.push_back(parse_inline(item)); continue; }
            ln.indent = inner; ln.text = trim(item);
            n.seq.push_back(parse_block(lines, i, inner));
        } else {
            if (ln.text.rfind("- ", 0) == 0) break;
            const size_t c = find_colon(ln.text);
            if (c == std::string::npos) throw Exception("expected 'key: value' but found: " + ln.text);
            const std::string key = unquote(trim(ln.text.substr(0, c)));
            const std::string rest = ln.text.substr(c + 1);
            ++i;
            n.map.emplace_back(key, parse_value_after_key(lines, i, indent, rest));
        }
    }
    if (i < lines.size() && lines[i].indent > indent) throw Exception("inconsistent indentation near: " + lines[i].text);
    return n;
}

} // namespace detail

inline Node Load(const std::string& text) {
    std::vector<detail::Line> lines;
    std::istringstream in(text);
    std::string raw;
    while (std::getline(in, raw)) {
        // strip comments (a '#' at line start or preceded by a space, outside quotes)
        bool sq = false, dq = false;
        for (size_t k = 0; k < raw.size(); ++k) {
            if (raw[k] == '\'' && !dq) sq = !sq;
            if (raw[k] == '"' && !sq) dq = !dq;
            if (raw[k] == '#' && !sq && !dq && (k == 0 || raw[k - 1] == ' ' || raw[k - 1] == '\t')) { raw.resize(k); break; }
        }
        const std::string t = detail::trim(raw);
        if (t.empty() || t == "---") continue;
        int ind = 0;
        while (ind < (int)raw.size() && raw[ind] == ' ') ++ind;
        lines.push_back(detail::Line{ind, t});
    }
    size_t i = 0;
    if (lines.empty()) return Node();
    if (lines.size() == 1 && (lines[0].text.front() == '{' || lines[0].text.front() == '[')) return detail::parse_inline(lines[0].text);
    Node root = detail::parse_block(lines, i, lines[0].indent);
    if (i != lines.size()) throw Exception("could not parse line: " + lines[i].text);
    return root;
}

} // namespace yaml
} // namespace nim
