// On-disk formats of the reference - SURVEY.md section 8f.4, host code by nature (file I/O):
//   se2lam::MapStorage        /root/reference/include/se2lam/MapStorage.h:28-97, src/MapStorage.cpp:31-603
//                             the map as ONE OpenCV FileStorage file (Config::WRITE_MAP_FILE_NAME = "se2lam.map": neither
//                             .xml nor .json, so cv::FileStorage writes YAML) plus one N.bmp per key frame
//   key-frame trajectory      /root/reference/src/OdoSLAM.cpp:198-212: "id x y z yaw" per key frame
//
// The reference delegates the byte layout to two un-vendored OpenCV 3.2 components:
//   cv::FileStorage (modules/core/src/persistence.cpp: icvYMLWrite, icvYMLStartWriteStruct / EndWriteStruct, icvFSFlush,
//       icvDoubleToString "%.16e", icvFloatToString "%.8e", icvWriteMat, wrap margin 71, block indent 3, flow indent +1,
//       "%YAML:1.0\n---\n" header, "...\n---\n" in front of every APPEND - MapStorage re-opens the file five times) and
//   cv::imwrite(".bmp") (modules/imgcodecs/src/grfmt_bmp.cpp: BmpEncoder::write - 8-bit image = 14 + 40 byte headers,
//       a 256-entry grey palette, rows bottom-up padded to 4 bytes).
// Both are restated below from their published sources; neither library is available in this image, so byte-parity with a
// file written by the real OpenCV is UNPINNED (like every other OpenCV restatement of this repository).  What the tests
// pin (tests/test_mapstorage.py, CPU): the emitter against an independent Python emitter of the same rules, the text
// parsed by a general YAML parser (PyYAML) back to the values that went in, write -> read -> write byte-identical, and the
// bitmap against an independent decoder.
//
// The pointer-graph classes (Map, KeyFrame, MapPoint) are outside the hot path; MapStorage works on the POD view
// `StoredMap` below, which holds exactly what MapStorage.cpp reads from / writes into them.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "types.h"

namespace se2lam_amd {

// ---------------------------------------------------------------------------------------------
// cv::Mat as the FileStorage sees it: 2-D, one channel, depth u (CV_8U) / i (CV_32S) / f (CV_32F) / d (CV_64F)
// ---------------------------------------------------------------------------------------------
struct StoredMat {
    int rows = 0, cols = 0;
    char dt = 'u';                 // a default-constructed cv::Mat has type 0 = CV_8UC1
    std::vector<uint8_t> u;
    std::vector<int32_t> i;
    std::vector<float> f;
    std::vector<double> d;
    size_t total() const { return (size_t)rows * cols; }
    static StoredMat zeros(int r, int c, char t) {
        StoredMat m;
        m.rows = r; m.cols = c; m.dt = t;
        const size_t n = (size_t)r * c;
        if (t == 'u') m.u.assign(n, 0); else if (t == 'i') m.i.assign(n, 0); else if (t == 'f') m.f.assign(n, 0.f); else m.d.assign(n, 0.0);
        return m;
    }
    bool operator==(const StoredMat& o) const {
        return rows == o.rows && cols == o.cols && dt == o.dt && u == o.u && i == o.i && f == o.f && d == o.d;
    }
};

// ---------------------------------------------------------------------------------------------
// The YAML emitter of cv::FileStorage (OpenCV 3.2 persistence.cpp), restated
// ---------------------------------------------------------------------------------------------
class CvYamlWriter {
public:
    enum { SEQ = 5, MAP = 6, TYPE_MASK = 7, FLOW = 8, EMPTY = 32 };   // CV_NODE_SEQ / MAP / FLOW / EMPTY
    // append = FileStorage::APPEND: a new YAML document behind what the file already holds
    explicit CvYamlWriter(bool append) { out_ = append ? "...\n---\n" : "%YAML:1.0\n---\n"; }

    void startStruct(const char* key, int flags, const char* type_name = nullptr) {   // icvYMLStartWriteStruct
        if (type_name && !*type_name) type_name = nullptr;
        flags = (flags & (TYPE_MASK | FLOW)) | EMPTY;
        std::string data;
        bool have = false;
        if (flags & FLOW) {
            const char c = (flags & TYPE_MASK) == MAP ? '{' : '[';
            data = type_name ? std::string("!!") + type_name + " " + c : std::string(1, c);
            have = true;
        } else if (type_name) {
            data = std::string("!!") + type_name;
            have = true;
        }
        write(key, have ? data.c_str() : nullptr);
        const int parent = flags_;
        stack_.push_back(parent);
        flags_ = flags;
        if (!(parent & FLOW)) indent_ += 3 + ((flags & FLOW) ? 1 : 0);
    }
    void endStruct() {                                                               // icvYMLEndWriteStruct
        const int flags = flags_;
        const int parent = stack_.back();
        stack_.pop_back();
        if (flags & FLOW) {
            if ((int)line_.size() > indent_ && !(flags & EMPTY)) line_ += ' ';
            line_ += (flags & TYPE_MASK) == MAP ? '}' : ']';
        } else if (flags & EMPTY) {
            flush();
            line_ += (flags & TYPE_MASK) == MAP ? "{}" : "[]";
        }
        if (!(parent & FLOW)) indent_ -= 3 + ((flags & FLOW) ? 1 : 0);
        flags_ = parent;
    }
    void writeInt(const char* key, int v) {                                          // icvYMLWriteInt
        char buf[32];
        std::snprintf(buf, sizeof buf, "%d", v);
        write(key, buf);
    }
    void writeReal(const char* key, double v) {                                      // icvYMLWriteReal
        char buf[64];
        write(key, doubleToString(buf, v));
    }
    void writeString(const char* key, const std::string& s) {                        // icvYMLWriteString, quote = 0
        const size_t len = s.size();
        if (!(len == 0 || s[0] != s[len - 1] || (s[0] != '\"' && s[0] != '\''))) { write(key, s.c_str()); return; }
        bool need_quote = len == 0 || s[0] == ' ';
        std::string b = "\"";
        for (char c : s) {
            const bool alnum = std::isalnum((unsigned char)c) != 0;
            if (!need_quote && !alnum && c != '_' && c != ' ' && c != '-' && c != '(' && c != ')' && c != '/' && c != '+' && c != ';')
                need_quote = true;
            if (!alnum && (!std::isprint((unsigned char)c) || c == '\\' || c == '\'' || c == '\"')) {
                b += '\\';
                if (std::isprint((unsigned char)c)) b += c;
                else if (c == '\n') b += 'n';
                else if (c == '\r') b += 'r';
                else if (c == '\t') b += 't';
                else { char h[8]; std::snprintf(h, sizeof h, "x%02x", (unsigned char)c); b += h; }
            } else {
                b += c;
            }
        }
        if (!need_quote && len && (std::isdigit((unsigned char)s[0]) || s[0] == '+' || s[0] == '-' || s[0] == '.')) need_quote = true;
        if (need_quote) b += '\"';
        write(key, need_quote ? b.c_str() : b.c_str() + 1);
    }
    // write(fs, name, Mat) -> cvWrite -> icvWriteMat: !!opencv-matrix { rows, cols, dt, data: [ flow ] }
    void writeMat(const char* key, const StoredMat& m) {
        startStruct(key, MAP, "opencv-matrix");
        writeInt("rows", m.rows);
        writeInt("cols", m.cols);
        writeString("dt", std::string(1, m.dt));
        startStruct("data", SEQ | FLOW);
        const size_t n = m.total();
        char buf[64];
        for (size_t k = 0; k < n; ++k) {                                             // cvWriteRawData, one scalar at a time
            if (m.dt == 'u') { std::snprintf(buf, sizeof buf, "%d", (int)m.u[k]); write(nullptr, buf); }
            else if (m.dt == 'i') { std::snprintf(buf, sizeof buf, "%d", m.i[k]); write(nullptr, buf); }
            else if (m.dt == 'f') write(nullptr, floatToString(buf, m.f[k]));
            else write(nullptr, doubleToString(buf, m.d[k]));
        }
        endStruct();
        endStruct();
    }
    // Point_<T> / Point3_<T>: WriteStructContext(fs, name, SEQ + FLOW) and the coordinates as scalars
    void writePoint2f(const char* key, float x, float y) { startStruct(key, SEQ | FLOW); writeReal(nullptr, x); writeReal(nullptr, y); endStruct(); }
    void writePoint3f(const char* key, float x, float y, float z) {
        startStruct(key, SEQ | FLOW); writeReal(nullptr, x); writeReal(nullptr, y); writeReal(nullptr, z); endStruct();
    }
    void writePoint2i(const char* key, int x, int y) { startStruct(key, SEQ | FLOW); writeInt(nullptr, x); writeInt(nullptr, y); endStruct(); }

    // FileStorage::release(): icvClose ends every structure still open (saveOdoGraph never writes its "]",
    // MapStorage.cpp:292-311) and flushes the last line
    std::string release() {
        while (!stack_.empty()) endStruct();
        flush();
        return out_;
    }

    static const char* doubleToString(char* buf, double value) {                     // icvDoubleToString
        if (std::isfinite(value)) {
            const int iv = cvRound(value);
            if ((double)iv == value) std::snprintf(buf, 64, "%d.", iv);
            else std::snprintf(buf, 64, "%.16e", value);
        } else if (std::isnan(value)) std::strcpy(buf, ".Nan");
        else std::strcpy(buf, value < 0 ? "-.Inf" : ".Inf");
        return buf;
    }
    static const char* floatToString(char* buf, float value) {                       // icvFloatToString
        if (std::isfinite(value)) {
            const int iv = cvRound((double)value);
            if ((float)iv == value) std::snprintf(buf, 64, "%d.", iv);
            else std::snprintf(buf, 64, "%.8e", (double)value);
        } else if (std::isnan(value)) std::strcpy(buf, ".Nan");
        else std::strcpy(buf, value < 0 ? "-.Inf" : ".Inf");
        return buf;
    }
    static int cvRound(double v) {   // round half to even (cvRound = lrint under the default rounding mode); out of range -> INT_MIN
        if (!(v > -2147483648.5 && v < 2147483647.5)) return (int)0x80000000;
        return (int)std::nearbyint(v);
    }

private:
    void flush() {                                                                   // icvFSFlush
        if ((int)line_.size() > space_) { out_ += line_; out_ += '\n'; }
        line_.assign((size_t)indent_, ' ');
        space_ = indent_;
    }
    void write(const char* key, const char* data) {                                  // icvYMLWrite
        if (key && !*key) key = nullptr;
        int flags = flags_;
        if ((flags & TYPE_MASK) != SEQ && (flags & TYPE_MASK) != MAP)
            flags = EMPTY | (key ? MAP : SEQ);                                       // top level: not yet a collection
        const size_t keylen = key ? std::strlen(key) : 0, datalen = data ? std::strlen(data) : 0;
        if (flags & FLOW) {
            if (!(flags & EMPTY)) line_ += ',';
            const int new_offset = (int)(line_.size() + keylen + datalen);
            if (new_offset > 71 && new_offset - indent_ > 10) flush();
            else line_ += ' ';
        } else {
            flush();
            if ((flags & TYPE_MASK) != MAP) {
                line_ += '-';
                if (data) line_ += ' ';
            }
        }
        if (key) {
            line_ += key;
            line_ += ':';
            if (!(flags & FLOW) && data) line_ += ' ';
        }
        if (data) line_ += data;
        flags_ = flags & ~EMPTY;
    }

    std::string out_, line_;
    std::vector<int> stack_;
    int flags_ = EMPTY, indent_ = 0, space_ = 0;
};

// ---------------------------------------------------------------------------------------------
// Reader for what the emitter above (= cv::FileStorage in YAML mode) produces: block maps and sequences, flow sequences
// and maps (multi-line), !!opencv-matrix maps (their data goes straight into a typed array), scalars, several documents
// in one file (FileStorage::operator[] looks a key up in every document, which is what makes APPEND work).
// ---------------------------------------------------------------------------------------------
struct YamlNode {
    enum Kind { NONE, SCALAR, SEQ, MAP, MAT } kind = NONE;
    std::string scalar;
    bool quoted = false;
    std::vector<YamlNode> seq;
    std::vector<std::pair<std::string, YamlNode>> map;
    StoredMat mat;
    const YamlNode* find(const std::string& key) const {
        for (const auto& kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const YamlNode& at(const std::string& key) const {
        const YamlNode* n = find(key);
        if (!n) throw std::runtime_error("MapStorage: key '" + key + "' is missing");
        return *n;
    }
    double real() const {
        if (kind != SCALAR) throw std::runtime_error("MapStorage: a scalar was expected");
        if (scalar == ".Inf" || scalar == "+.Inf") return HUGE_VAL;
        if (scalar == "-.Inf") return -HUGE_VAL;
        if (scalar == ".Nan" || scalar == ".NaN" || scalar == ".nan") return std::nan("");
        char* end = nullptr;
        const double v = std::strtod(scalar.c_str(), &end);
        if (end == scalar.c_str() || *end) throw std::runtime_error("MapStorage: '" + scalar + "' is not a number");
        return v;
    }
    int integer() const { return CvYamlWriter::cvRound(real()); }   // (int)FileNode of a real node rounds, as cv::FileNode does
};

class CvYamlReader {
public:
    explicit CvYamlReader(const std::string& text) : s_(text) { parseDocuments(); }
    // FileStorage::operator[]: the first document that has the key
    const YamlNode& operator[](const std::string& key) const {
        for (const auto& d : docs_)
            if (const YamlNode* n = d.find(key)) return *n;
        throw std::runtime_error("MapStorage: the file has no node '" + key + "'");
    }
    bool has(const std::string& key) const {
        for (const auto& d : docs_)
            if (d.find(key)) return true;
        return false;
    }
    size_t documents() const { return docs_.size(); }

private:
    [[noreturn]] void fail(const std::string& what) const {
        size_t line = 1;
        for (size_t k = 0; k < pos_ && k < s_.size(); ++k) line += s_[k] == '\n';
        throw std::runtime_error("MapStorage: YAML line " + std::to_string(line) + ": " + what);
    }
    bool eof() const { return pos_ >= s_.size(); }
    size_t lineEnd(size_t p) const { const size_t e = s_.find('\n', p); return e == std::string::npos ? s_.size() : e; }
    // position at the first non-blank line at or after pos_; returns its indentation, -1 at the end of the text
    int peekIndent() {
        for (;;) {
            if (eof()) return -1;
            size_t p = pos_;
            while (p < s_.size() && s_[p] == ' ') ++p;
            if (p >= s_.size()) { pos_ = p; return -1; }
            if (s_[p] == '\n' || s_[p] == '\r' || s_[p] == '#') { pos_ = lineEnd(p) + 1; continue; }
            return (int)(p - pos_);
        }
    }
    bool startsWith(size_t p, const char* lit) const { return s_.compare(p, std::strlen(lit), lit) == 0; }

    void parseDocuments() {
        pos_ = 0;
        if (startsWith(0, "%YAML")) pos_ = lineEnd(0) + 1;
        YamlNode doc;
        doc.kind = YamlNode::MAP;
        bool any = false;
        for (;;) {
            const int ind = peekIndent();
            if (ind < 0) break;
            const size_t p = pos_ + (size_t)ind;
            if (ind == 0 && (startsWith(p, "---") || startsWith(p, "..."))) {
                if (startsWith(p, "---") && any) { docs_.push_back(std::move(doc)); doc = YamlNode(); doc.kind = YamlNode::MAP; any = false; }
                pos_ = lineEnd(p) + 1;
                continue;
            }
            if (ind != 0) fail("a top-level key was expected");
            YamlNode m = parseBlock(0);
            if (m.kind != YamlNode::MAP) fail("the top level of a document must be a map");
            for (auto& kv : m.map) doc.map.push_back(std::move(kv));
            any = true;
        }
        if (any || docs_.empty()) docs_.push_back(std::move(doc));
    }

    // a block collection whose entries start at column `indent` (pos_ is at the start of its first line)
    YamlNode parseBlock(int indent) {
        YamlNode node;
        const size_t p0 = pos_ + (size_t)indent;
        if (s_[p0] == '[' || s_[p0] == '{') {          // "[]" / "{}" of an empty block structure, on a line of its own
            pos_ = p0;
            node = parseFlow();
            pos_ = lineEnd(pos_) + 1;
            return node;
        }
        const bool is_seq = s_[p0] == '-' && (p0 + 1 >= s_.size() || s_[p0 + 1] == ' ' || s_[p0 + 1] == '\n' || s_[p0 + 1] == '\r');
        node.kind = is_seq ? YamlNode::SEQ : YamlNode::MAP;
        for (;;) {
            const int ind = peekIndent();
            if (ind != indent) {
                if (ind > indent) fail("unexpected indentation");
                break;
            }
            size_t p = pos_ + (size_t)indent;
            if (indent == 0 && (startsWith(p, "---") || startsWith(p, "..."))) break;
            if (is_seq) {
                if (s_[p] != '-') break;
                ++p;
                while (p < s_.size() && s_[p] == ' ') ++p;
                node.seq.push_back(parseValueAt(p, indent));
            } else {
                if (s_[p] == '-') fail("a sequence entry inside a map");
                const size_t e = lineEnd(p);
                size_t c = p;
                while (c < e && s_[c] != ':') ++c;
                if (c >= e) fail("'key:' expected");
                std::string key = s_.substr(p, c - p);
                while (!key.empty() && key.back() == ' ') key.pop_back();
                p = c + 1;
                while (p < e && s_[p] == ' ') ++p;
                node.map.emplace_back(std::move(key), parseValueAt(p, indent));
            }
        }
        return node;
    }

    // the value that starts at p on the current line (possibly nothing: the value is the deeper block that follows);
    // leaves pos_ at the start of the first line behind the value
    YamlNode parseValueAt(size_t p, int parent_indent) {
        size_t e = lineEnd(p);
        std::string tag;
        if (p < e && s_[p] == '!') {
            size_t t = p;
            while (t < e && s_[t] != ' ' && s_[t] != '\r') ++t;
            tag = s_.substr(p, t - p);
            p = t;
            while (p < e && s_[p] == ' ') ++p;
        }
        while (e > p && (s_[e - 1] == ' ' || s_[e - 1] == '\r')) --e;
        if (p >= e) {                                    // nothing on this line: a nested block (or an empty value)
            pos_ = lineEnd(p) + 1;
            const int ind = peekIndent();
            if (ind <= parent_indent) { YamlNode n; n.kind = YamlNode::SCALAR; return n; }
            if (tag == "!!opencv-matrix") return parseMatrix(ind);
            return parseBlock(ind);
        }
        if (s_[p] == '[' || s_[p] == '{') {
            pos_ = p;
            YamlNode n = parseFlow();
            pos_ = lineEnd(pos_) + 1;
            return n;
        }
        YamlNode n = scalarOf(p, e);
        pos_ = lineEnd(p) + 1;
        return n;
    }

    YamlNode scalarOf(size_t p, size_t e) const {
        YamlNode n;
        n.kind = YamlNode::SCALAR;
        if (e - p >= 2 && (s_[p] == '"' || s_[p] == '\'') && s_[e - 1] == s_[p]) {
            n.quoted = true;
            for (size_t k = p + 1; k + 1 < e; ++k) {
                char c = s_[k];
                if (c == '\\' && s_[p] == '"' && k + 2 < e) {
                    c = s_[++k];
                    if (c == 'n') c = '\n'; else if (c == 'r') c = '\r'; else if (c == 't') c = '\t';
                    else if (c == 'x' && k + 3 < e) { c = (char)std::strtol(s_.substr(k + 1, 2).c_str(), nullptr, 16); k += 2; }
                }
                n.scalar += c;
            }
        } else {
            n.scalar = s_.substr(p, e - p);
        }
        return n;
    }

    void skipFlowSpace() {
        while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\n' || s_[pos_] == '\r' || s_[pos_] == '\t')) ++pos_;
    }
    // '[' ... ']' or '{' ... '}' starting at pos_, over as many lines as it takes; leaves pos_ behind the closing bracket
    YamlNode parseFlow() {
        YamlNode node;
        const char open = s_[pos_++];
        const char close = open == '[' ? ']' : '}';
        node.kind = open == '[' ? YamlNode::SEQ : YamlNode::MAP;
        for (;;) {
            skipFlowSpace();
            if (eof()) fail("unterminated flow collection");
            if (s_[pos_] == close) { ++pos_; break; }
            if (s_[pos_] == ',') { ++pos_; continue; }
            std::string key;
            if (open == '{') {
                const size_t k0 = pos_;
                while (pos_ < s_.size() && s_[pos_] != ':' && s_[pos_] != '}' && s_[pos_] != '\n') ++pos_;
                if (eof() || s_[pos_] != ':') fail("'key:' expected in a flow map");
                key = s_.substr(k0, pos_ - k0);
                while (!key.empty() && key.back() == ' ') key.pop_back();
                ++pos_;
                skipFlowSpace();
            }
            YamlNode v;
            if (s_[pos_] == '[' || s_[pos_] == '{') v = parseFlow();
            else {
                const size_t v0 = pos_;
                if (s_[pos_] == '"' || s_[pos_] == '\'') {
                    const char q = s_[pos_++];
                    while (pos_ < s_.size() && s_[pos_] != q) pos_ += (s_[pos_] == '\\' && q == '"') ? 2 : 1;
                    if (eof()) fail("unterminated string");
                    ++pos_;
                } else {
                    while (pos_ < s_.size() && s_[pos_] != ',' && s_[pos_] != close && s_[pos_] != '\n') ++pos_;
                }
                size_t v1 = pos_;
                while (v1 > v0 && (s_[v1 - 1] == ' ' || s_[v1 - 1] == '\r')) --v1;
                v = scalarOf(v0, v1);
            }
            if (open == '{') node.map.emplace_back(std::move(key), std::move(v));
            else node.seq.push_back(std::move(v));
        }
        return node;
    }

    // the block map behind "!!opencv-matrix": rows, cols, dt, data: [ ... ] - data parsed straight into the typed array
    YamlNode parseMatrix(int indent) {
        YamlNode node;
        node.kind = YamlNode::MAT;
        StoredMat& m = node.mat;
        bool have_dt = false;
        for (;;) {
            const int ind = peekIndent();
            if (ind != indent) break;
            size_t p = pos_ + (size_t)indent;
            const size_t e = lineEnd(p);
            size_t c = p;
            while (c < e && s_[c] != ':') ++c;
            if (c >= e) fail("'key:' expected inside an opencv-matrix");
            const std::string key = s_.substr(p, c - p);
            p = c + 1;
            while (p < e && s_[p] == ' ') ++p;
            if (key == "data") {
                if (!have_dt) fail("opencv-matrix: dt must precede data");
                if (p >= e || s_[p] != '[') fail("opencv-matrix: data must be a flow sequence");
                pos_ = p + 1;
                const size_t n = m.total();
                if (m.dt == 'u') m.u.reserve(n); else if (m.dt == 'i') m.i.reserve(n); else if (m.dt == 'f') m.f.reserve(n); else m.d.reserve(n);
                size_t got = 0;
                for (;;) {
                    skipFlowSpace();
                    if (eof()) fail("opencv-matrix: unterminated data");
                    if (s_[pos_] == ']') { ++pos_; break; }
                    if (s_[pos_] == ',') { ++pos_; continue; }
                    const size_t v0 = pos_;
                    while (pos_ < s_.size() && s_[pos_] != ',' && s_[pos_] != ']' && s_[pos_] != '\n' && s_[pos_] != ' ') ++pos_;
                    YamlNode v;
                    v.kind = YamlNode::SCALAR;
                    v.scalar = s_.substr(v0, pos_ - v0);
                    const double x = v.real();
                    if (m.dt == 'u') m.u.push_back((uint8_t)x); else if (m.dt == 'i') m.i.push_back((int32_t)x);
                    else if (m.dt == 'f') m.f.push_back((float)x); else m.d.push_back(x);
                    ++got;
                }
                if (got != n) fail("opencv-matrix: " + std::to_string(got) + " values for " + std::to_string(m.rows) + " x " + std::to_string(m.cols));
                pos_ = lineEnd(pos_) + 1;
            } else {
                YamlNode v = parseValueAt(p, indent);
                if (key == "rows") m.rows = v.integer();
                else if (key == "cols") m.cols = v.integer();
                else if (key == "dt") {
                    if (v.scalar.size() != 1 || !std::strchr("uifd", v.scalar[0])) fail("opencv-matrix: dt '" + v.scalar + "' is not one of u, i, f, d");
                    m.dt = v.scalar[0];
                    have_dt = true;
                }
            }
        }
        return node;
    }

    const std::string& s_;
    size_t pos_ = 0;
    std::vector<YamlNode> docs_;
};

// ---------------------------------------------------------------------------------------------
// cv::imwrite(".bmp") of a CV_8UC1 image / cv::imread(.., CV_LOAD_IMAGE_GRAYSCALE) (grfmt_bmp.cpp), restated
// ---------------------------------------------------------------------------------------------
inline std::string encodeBmpGray(const uint8_t* img, int rows, int cols, size_t step) {
    const int fileStep = (cols + 3) & -4, headerSize = 14 + 40 + 1024;
    const uint32_t fileSize = (uint32_t)((size_t)fileStep * rows + headerSize);
    std::string o;
    auto dw = [&](uint32_t v) { for (int k = 0; k < 4; ++k) o += (char)((v >> (8 * k)) & 255); };
    auto w = [&](uint16_t v) { o += (char)(v & 255); o += (char)(v >> 8); };
    o += "BM";
    dw(fileSize); dw(0); dw((uint32_t)headerSize);
    dw(40); dw((uint32_t)cols); dw((uint32_t)rows); w(1); w(8); dw(0 /*BMP_RGB*/); dw(0); dw(0); dw(0); dw(0); dw(0);
    for (int k = 0; k < 256; ++k) { o += (char)k; o += (char)k; o += (char)k; o += (char)0; }   // FillGrayPalette
    for (int y = rows - 1; y >= 0; --y) {
        o.append(reinterpret_cast<const char*>(img + (size_t)y * step), (size_t)cols);
        o.append((size_t)(fileStep - cols), '\0');
    }
    return o;
}
// 8-bit paletted (what imwrite produces) and 24 / 32-bit uncompressed files, bottom-up or top-down, to grey
inline bool decodeBmpGray(const std::string& d, Mat8U& out) {
    auto rd = [&](size_t off, int n) { uint32_t v = 0; for (int k = 0; k < n; ++k) v |= (uint32_t)(uint8_t)d[off + k] << (8 * k); return v; };
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return false;
    const uint32_t offset = rd(10, 4), hsize = rd(14, 4);
    if (hsize < 40) return false;
    const int width = (int)rd(18, 4);
    int height = (int)rd(22, 4);
    const int bpp = (int)rd(28, 2);
    const uint32_t compression = rd(30, 4);
    if (compression != 0 || width <= 0 || height == 0 || (bpp != 8 && bpp != 24 && bpp != 32)) return false;
    const bool top_down = height < 0;
    if (top_down) height = -height;
    uint32_t ncol = rd(46, 4);
    if (bpp == 8 && ncol == 0) ncol = 256;
    uint8_t gray[256];
    for (int k = 0; k < 256; ++k) gray[k] = (uint8_t)k;
    if (bpp == 8) {
        const size_t pal = 14 + (size_t)hsize;
        if (pal + 4 * (size_t)ncol > d.size()) return false;
        for (uint32_t k = 0; k < ncol && k < 256; ++k) {      // icvCvt_BGR2Gray_8u_C3C1R: (b 1868 + g 9617 + r 4899 + 8192) >> 14
            const unsigned b = (uint8_t)d[pal + 4 * k], g = (uint8_t)d[pal + 4 * k + 1], r = (uint8_t)d[pal + 4 * k + 2];
            gray[k] = (uint8_t)((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14);
        }
    }
    const size_t fileStep = (((size_t)width * (bpp / 8)) + 3) & ~(size_t)3;
    if ((size_t)offset + fileStep * (size_t)height > d.size()) return false;
    out.create(height, width);
    for (int y = 0; y < height; ++y) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(d.data()) + offset + fileStep * (size_t)(top_down ? y : height - 1 - y);
        uint8_t* dst = out.ptr(y);
        if (bpp == 8) for (int x = 0; x < width; ++x) dst[x] = gray[src[x]];
        else {
            const int cn = bpp / 8;
            for (int x = 0; x < width; ++x)
                dst[x] = (uint8_t)((src[cn * x] * 1868u + src[cn * x + 1] * 9617u + src[cn * x + 2] * 4899u + 8192u) >> 14);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// The map as MapStorage sees it
// ---------------------------------------------------------------------------------------------
struct StoredFtrEdge {           // KeyFrame::mFtrMeasureFrom entry: target key frame (index into kfs) + SE3Constraint
    int to = -1;
    StoredMat measure, info;
};
struct StoredKeyFrame {
    int id = 0;                  // KeyFrame::id (the frame id: what the trajectory file prints; not stored in the map file)
    int mIdKF = 0;               // renumbered to the vector index by saveMap (sortKeyFrames)
    bool null = false;           // KeyFrame::isNull(): dropped by saveMap
    std::vector<KeyPoint> keyPoints, keyPointsUn;
    StoredMat descriptors;       // N x 32 'u'
    std::vector<Point3f> mViewMPs;
    std::vector<Matrix3D> mViewMPsInfo;    // Eigen::Matrix3d each; written through toCvMat = CV_32F (converter.cpp:110-118)
    MatF Tcw = MatF::eye(4);     // getPose()
    float odom[3] = {0, 0, 0};   // Se2 odom (x, y, theta)
    float mfScaleFactor = 1.2f;
    Mat8U img;                   // owned; N.bmp
    std::vector<std::pair<int, int>> observations;    // (map point index into mps, feature index): hasObservation / getFtrIdx
    std::vector<int> covisible;                       // getAllCovisibleKFs() as indices into kfs
    int odoNext = -1;                                 // mOdoMeasureFrom.first (index into kfs), -1 = none
    StoredMat odoMeasure, odoInfo;                    // mOdoMeasureFrom.second.measure / .info
    std::vector<StoredFtrEdge> ftrFrom;               // mFtrMeasureFrom
};
struct StoredMapPoint {
    int mId = 0;
    bool null = false, goodPrl = true;    // saveMap keeps !isNull() && isGoodPrl() (sortMapPoints); loadMap sets goodPrl
    Point3f pos;
};
struct StoredMap {
    std::vector<StoredKeyFrame> kfs;
    std::vector<StoredMapPoint> mps;
    void clear() { kfs.clear(); mps.clear(); }
};

class MapStorage {
public:
    MapStorage() = default;
    void setMap(StoredMap* pMap) { mpMap = pMap; }                                                   // MapStorage.cpp:28
    void setFilePath(const std::string path, const std::string file) { mMapPath = path; mMapFile = file; }   // :23
    void clearData() { kf_of_.clear(); mp_of_.clear(); }

    // MapStorage::saveMap (:53-75): null key frames and null / bad-parallax map points dropped, ids = vector indices, then
    // the six sections - "KeyFrames" with WRITE, the other five each with its own APPEND
    void saveMap() {
        if (!mpMap) throw std::runtime_error("MapStorage::saveMap: setMap first");
        sortKeyFrames();
        sortMapPoints();
        std::string text = saveKeyFrames();
        text += saveMapPoints();
        text += saveObservations();
        text += saveCovisibilityGraph();
        text += saveOdoGraph();
        text += saveFtrGraph();
        writeFile(mMapPath + mMapFile, text);
    }

    // MapStorage::loadMap (:31-51): the map is cleared and rebuilt from the file and the N.bmp images
    void loadMap() {
        if (!mpMap) throw std::runtime_error("MapStorage::loadMap: setMap first");
        const std::string text = readFile(mMapPath + mMapFile);
        CvYamlReader fs(text);
        StoredMap m;
        loadKeyFrames(fs, m);
        loadMapPoints(fs, m);
        loadObservations(fs, m);
        loadCovisibilityGraph(fs, m);
        loadOdoGraph(fs, m);
        loadFtrGraph(fs, m);
        *mpMap = std::move(m);                                                                        // loadToMap (:583-591)
    }

    // the YAML text of the last saveMap / what loadMap parses, without touching the disk (tests)
    std::string mapText() {
        sortKeyFrames();
        sortMapPoints();
        return saveKeyFrames(false) + saveMapPoints() + saveObservations() + saveCovisibilityGraph() + saveOdoGraph() + saveFtrGraph();
    }

private:
    static void writeFile(const std::string& path, const std::string& data) {
        std::ofstream f(path, std::ios::binary | std::ios::trunc);
        if (!f) throw std::runtime_error("MapStorage: cannot write " + path);
        f.write(data.data(), (std::streamsize)data.size());
        if (!f) throw std::runtime_error("MapStorage: short write to " + path);
    }
    static std::string readFile(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("MapStorage: cannot read " + path);
        std::ostringstream ss;
        ss << f.rdbuf();
        return ss.str();
    }

    void sortKeyFrames() {                                   // :77-98
        kf_of_.clear();
        old_of_kf_.assign(mpMap->kfs.size(), -1);
        for (int i = 0; i < (int)mpMap->kfs.size(); ++i)
            if (!mpMap->kfs[i].null) { old_of_kf_[i] = (int)kf_of_.size(); kf_of_.push_back(i); }
        for (int i = 0; i < (int)kf_of_.size(); ++i) mpMap->kfs[kf_of_[i]].mIdKF = i;
    }
    void sortMapPoints() {                                   // :100-120
        mp_of_.clear();
        old_of_mp_.assign(mpMap->mps.size(), -1);
        for (int i = 0; i < (int)mpMap->mps.size(); ++i)
            if (!mpMap->mps[i].null && mpMap->mps[i].goodPrl) { old_of_mp_[i] = (int)mp_of_.size(); mp_of_.push_back(i); }
        for (int i = 0; i < (int)mp_of_.size(); ++i) mpMap->mps[mp_of_[i]].mId = i;
    }

    static void writeKeyPoints(CvYamlWriter& file, const char* name, const std::vector<KeyPoint>& kps) {
        file.startStruct(name, CvYamlWriter::SEQ);
        for (const KeyPoint& kp : kps) {
            file.startStruct(nullptr, CvYamlWriter::MAP);
            file.writePoint2f("pt", kp.pt.x, kp.pt.y);
            file.writeInt("octave", kp.octave);
            file.writeReal("angle", kp.angle);
            file.writeReal("response", kp.response);
            file.endStruct();
        }
        file.endStruct();
    }
    static StoredMat matOf(const MatF& m) {
        StoredMat s = StoredMat::zeros(m.rows, m.cols, 'f');
        s.f = m.v;
        return s;
    }

    std::string saveKeyFrames(bool images = true) {          // :122-198
        if (images)
            for (int i = 0; i < (int)kf_of_.size(); ++i) {
                const StoredKeyFrame& kf = mpMap->kfs[kf_of_[i]];
                writeFile(mMapPath + std::to_string(i) + ".bmp", encodeBmpGray(kf.img.data, kf.img.rows, kf.img.cols, kf.img.step));
            }
        CvYamlWriter file(false);
        file.startStruct("KeyFrames", CvYamlWriter::SEQ);
        for (int i = 0; i < (int)kf_of_.size(); ++i) {
            const StoredKeyFrame& kf = mpMap->kfs[kf_of_[i]];
            file.startStruct(nullptr, CvYamlWriter::MAP);
            file.writeInt("Id", i);
            writeKeyPoints(file, "KeyPoints", kf.keyPoints);
            writeKeyPoints(file, "KeyPointsUn", kf.keyPointsUn);
            file.writeMat("Descriptor", kf.descriptors);
            file.startStruct("ViewMPs", CvYamlWriter::SEQ);
            for (const Point3f& p : kf.mViewMPs) file.writePoint3f(nullptr, p.x, p.y, p.z);
            file.endStruct();
            file.startStruct("ViewMPInfo", CvYamlWriter::SEQ);
            for (const Matrix3D& I : kf.mViewMPsInfo) {
                StoredMat s = StoredMat::zeros(3, 3, 'f');
                for (int k = 0; k < 9; ++k) s.f[k] = (float)I.m[k];
                file.writeMat(nullptr, s);
            }
            file.endStruct();
            file.writeMat("Pose", matOf(kf.Tcw));
            file.writePoint3f("Odometry", kf.odom[0], kf.odom[1], kf.odom[2]);
            file.writeReal("ScaleFactor", kf.mfScaleFactor);
            file.endStruct();
        }
        file.endStruct();
        return file.release();
    }
    std::string saveMapPoints() {                            // :200-220
        CvYamlWriter file(true);
        file.startStruct("MapPoints", CvYamlWriter::SEQ);
        for (int i = 0; i < (int)mp_of_.size(); ++i) {
            const StoredMapPoint& mp = mpMap->mps[mp_of_[i]];
            file.startStruct(nullptr, CvYamlWriter::MAP);
            file.writeInt("Id", i);
            file.writePoint3f("Pos", mp.pos.x, mp.pos.y, mp.pos.z);
            file.endStruct();
        }
        file.endStruct();
        return file.release();
    }
    std::string saveObservations() {                         // :222-253: dense sizeKF x sizeMP matrices
        const int nk = (int)kf_of_.size(), nm = (int)mp_of_.size();
        StoredMat obs = StoredMat::zeros(nk, nm, 'i'), index = StoredMat::zeros(nk, nm, 'i');
        std::fill(index.i.begin(), index.i.end(), -1);
        for (int i = 0; i < nk; ++i)
            for (const auto& ob : mpMap->kfs[kf_of_[i]].observations) {
                const int j = ob.first >= 0 && ob.first < (int)old_of_mp_.size() ? old_of_mp_[ob.first] : -1;
                if (j < 0) continue;
                obs.i[(size_t)i * nm + j] = 1;
                index.i[(size_t)i * nm + j] = ob.second;
            }
        CvYamlWriter file(true);
        file.writeMat("Observations", obs);
        file.writeMat("ObservationIndex", index);
        return file.release();
    }
    std::string saveCovisibilityGraph() {                    // :255-274
        const int nk = (int)kf_of_.size();
        StoredMat g = StoredMat::zeros(nk, nk, 'i');
        for (int i = 0; i < nk; ++i)
            for (int c : mpMap->kfs[kf_of_[i]].covisible) {
                const int j = c >= 0 && c < (int)old_of_kf_.size() ? old_of_kf_[c] : -1;
                if (j >= 0) g.i[(size_t)i * nk + j] = 1;
            }
        CvYamlWriter file(true);
        file.writeMat("CovisibilityGraph", g);
        return file.release();
    }
    std::string saveOdoGraph() {                             // :276-311 (the sequence is closed by release(), not by "]")
        CvYamlWriter file(true);
        file.startStruct("OdoGraphNextKF", CvYamlWriter::SEQ);
        for (int i = 0; i < (int)kf_of_.size(); ++i) {
            const StoredKeyFrame& kf = mpMap->kfs[kf_of_[i]];
            const int next = kf.odoNext >= 0 && kf.odoNext < (int)old_of_kf_.size() ? old_of_kf_[kf.odoNext] : -1;
            file.startStruct(nullptr, CvYamlWriter::MAP);
            file.writeInt("NextId", next);
            file.writeMat("Measure", kf.odoMeasure);
            file.writeMat("Info", kf.odoInfo);
            file.endStruct();
        }
        return file.release();
    }
    std::string saveFtrGraph() {                             // :313-346
        CvYamlWriter file(true);
        file.startStruct("FtrGraphPairs", CvYamlWriter::SEQ);
        for (int i = 0; i < (int)kf_of_.size(); ++i)
            for (const StoredFtrEdge& e : mpMap->kfs[kf_of_[i]].ftrFrom) {
                const int j = e.to >= 0 && e.to < (int)old_of_kf_.size() ? old_of_kf_[e.to] : -1;
                if (j < 0) continue;
                file.startStruct(nullptr, CvYamlWriter::MAP);
                file.writePoint2i("PairId", i, j);
                file.writeMat("Measure", e.measure);
                file.writeMat("Info", e.info);
                file.endStruct();
            }
        file.endStruct();
        return file.release();
    }

    static const StoredMat& matNode(const YamlNode& n) {
        if (n.kind != YamlNode::MAT) throw std::runtime_error("MapStorage: an opencv-matrix was expected");
        return n.mat;
    }
    static void readKeyPoints(const YamlNode& seq, std::vector<KeyPoint>& out) {
        out.clear();
        for (const YamlNode& n : seq.seq) {
            KeyPoint kp;                                    // cv::KeyPoint(): size 0, angle -1, response 0, octave 0, class_id -1
            const YamlNode& pt = n.at("pt");
            if (pt.seq.size() != 2) throw std::runtime_error("MapStorage: pt needs two coordinates");
            kp.pt.x = (float)pt.seq[0].real(); kp.pt.y = (float)pt.seq[1].real();
            kp.octave = n.at("octave").integer();
            kp.angle = (float)n.at("angle").real();
            kp.response = (float)n.at("response").real();
            out.push_back(kp);
        }
    }
    static Point3f point3(const YamlNode& n) {
        if (n.seq.size() != 3) throw std::runtime_error("MapStorage: a 3-D point needs three coordinates");
        Point3f p;
        p.x = (float)n.seq[0].real(); p.y = (float)n.seq[1].real(); p.z = (float)n.seq[2].real();
        return p;
    }

    void loadKeyFrames(const CvYamlReader& fs, StoredMap& m) {       // :348-446
        const YamlNode& nodeKFs = fs["KeyFrames"];
        for (const YamlNode& n : nodeKFs.seq) {
            StoredKeyFrame kf;
            kf.mIdKF = n.at("Id").integer();
            readKeyPoints(n.at("KeyPoints"), kf.keyPoints);
            readKeyPoints(n.at("KeyPointsUn"), kf.keyPointsUn);
            kf.descriptors = matNode(n.at("Descriptor"));
            for (const YamlNode& p : n.at("ViewMPs").seq) kf.mViewMPs.push_back(point3(p));
            for (const YamlNode& I : n.at("ViewMPInfo").seq) {
                const StoredMat& s = matNode(I);
                if (s.total() != 9 || s.dt != 'f') throw std::runtime_error("MapStorage: ViewMPInfo entries are 3x3 CV_32F");
                Matrix3D M;
                for (int k = 0; k < 9; ++k) M.m[k] = (double)s.f[k];                                  // toMatrix3d
                kf.mViewMPsInfo.push_back(M);
            }
            const StoredMat& pose = matNode(n.at("Pose"));
            if (pose.rows != 4 || pose.cols != 4 || pose.dt != 'f') throw std::runtime_error("MapStorage: Pose is 4x4 CV_32F");
            kf.Tcw = MatF(4, 4);
            kf.Tcw.v = pose.f;
            const Point3f odo = point3(n.at("Odometry"));
            kf.odom[0] = odo.x; kf.odom[1] = odo.y; kf.odom[2] = odo.z;
            kf.mfScaleFactor = (float)n.at("ScaleFactor").real();
            m.kfs.push_back(std::move(kf));
        }
        for (int i = 0; i < (int)m.kfs.size(); ++i) {      // imread(mMapPath + to_string(i) + ".bmp", CV_LOAD_IMAGE_GRAYSCALE)
            std::ifstream f(mMapPath + std::to_string(i) + ".bmp", std::ios::binary);
            if (!f) continue;                               // imread returns an empty Mat; the key frame keeps an empty image
            std::ostringstream ss;
            ss << f.rdbuf();
            if (!decodeBmpGray(ss.str(), m.kfs[i].img)) m.kfs[i].img.release();
        }
    }
    void loadMapPoints(const CvYamlReader& fs, StoredMap& m) {       // :448-475
        for (const YamlNode& n : fs["MapPoints"].seq) {
            StoredMapPoint mp;
            mp.mId = n.at("Id").integer();
            mp.pos = point3(n.at("Pos"));
            mp.goodPrl = true;
            m.mps.push_back(mp);
        }
    }
    void loadObservations(const CvYamlReader& fs, StoredMap& m) {    // :477-506
        const StoredMat& obs = matNode(fs["Observations"]);
        const StoredMat& index = matNode(fs["ObservationIndex"]);
        if (obs.dt != 'i' || index.dt != 'i' || obs.rows != index.rows || obs.cols != index.cols)
            throw std::runtime_error("MapStorage: Observations / ObservationIndex are CV_32S matrices of one size");
        if (obs.rows > (int)m.kfs.size() || obs.cols > (int)m.mps.size()) throw std::runtime_error("MapStorage: Observations exceed the map");
        for (int i = 0; i < obs.rows; ++i)
            for (int j = 0; j < obs.cols; ++j)
                if (obs.i[(size_t)i * obs.cols + j]) m.kfs[i].observations.emplace_back(j, index.i[(size_t)i * obs.cols + j]);
    }
    void loadCovisibilityGraph(const CvYamlReader& fs, StoredMap& m) {   // :508-532: added both ways (std::set semantics)
        const StoredMat& g = matNode(fs["CovisibilityGraph"]);
        if (g.dt != 'i' || g.rows > (int)m.kfs.size() || g.cols > (int)m.kfs.size()) throw std::runtime_error("MapStorage: bad CovisibilityGraph");
        auto add = [](std::vector<int>& v, int x) {
            for (int y : v) if (y == x) return;
            v.push_back(x);
        };
        for (int i = 0; i < g.rows; ++i)
            for (int j = 0; j < g.cols; ++j)
                if (g.i[(size_t)i * g.cols + j]) { add(m.kfs[i].covisible, j); add(m.kfs[j].covisible, i); }
    }
    void loadOdoGraph(const CvYamlReader& fs, StoredMap& m) {         // :534-564
        const YamlNode& nodes = fs["OdoGraphNextKF"];
        for (int i = 0; i < (int)nodes.seq.size() && i < (int)m.kfs.size(); ++i) {
            const YamlNode& n = nodes.seq[i];
            const int j = n.at("NextId").integer();
            if (j < 0) continue;
            if (j >= (int)m.kfs.size()) throw std::runtime_error("MapStorage: OdoGraphNextKF names a key frame that does not exist");
            m.kfs[i].odoNext = j;
            m.kfs[i].odoMeasure = matNode(n.at("Measure"));
            m.kfs[i].odoInfo = matNode(n.at("Info"));
        }
    }
    void loadFtrGraph(const CvYamlReader& fs, StoredMap& m) {         // :566-581
        for (const YamlNode& n : fs["FtrGraphPairs"].seq) {
            const YamlNode& pid = n.at("PairId");
            if (pid.seq.size() != 2) throw std::runtime_error("MapStorage: PairId needs two ids");
            const int a = pid.seq[0].integer(), b = pid.seq[1].integer();
            if (a < 0 || b < 0 || a >= (int)m.kfs.size() || b >= (int)m.kfs.size()) throw std::runtime_error("MapStorage: FtrGraphPairs names a key frame that does not exist");
            // KeyFrame::addFtrMeasureFrom is std::map::insert (KeyFrame.cpp:221-223): a second constraint between the same pair of key
            // frames is ignored, the first stays - seen running in the compiled reference (tests/test_ref_compiled.py)
            bool known = false;
            for (const StoredFtrEdge& have : m.kfs[a].ftrFrom) known = known || have.to == b;
            if (known) continue;
            StoredFtrEdge e;
            e.to = b;
            e.measure = matNode(n.at("Measure"));
            e.info = matNode(n.at("Info"));
            m.kfs[a].ftrFrom.push_back(std::move(e));
        }
    }

    StoredMap* mpMap = nullptr;
    std::string mMapPath, mMapFile;
    std::vector<int> kf_of_, mp_of_, old_of_kf_, old_of_mp_;   // stored index -> map index and back (-1 = dropped)
};

// ---------------------------------------------------------------------------------------------
// OdoSLAM::saveMap's key-frame trajectory (OdoSLAM.cpp:198-212): per non-null key frame
//     id  wTb(0,3)  wTb(1,3)  wTb(2,3)  yaw        with wTb = cvu::inv(Config::bTc * Tcw)  (CV_32F arithmetic)
// yaw = g2o::internal::toEuler(wRb)(2) [3P g2o 20160424, types/slam3d/isometry3d_mappings.cpp: Quaterniond(R), then
// atan2(2 (q0 q3 + q1 q2), 1 - 2 (q2^2 + q3^2))].  Numbers as std::ostream prints them (6 significant digits, %g).
// ---------------------------------------------------------------------------------------------
struct TrajectoryEntry {
    int id = 0;
    float x = 0, y = 0, z = 0;
    double yaw = 0;
};
inline TrajectoryEntry trajectoryEntry(int id, const MatF& bTc, const MatF& Tcw) {
    float T[16];                                             // bTc * Tcw: cv::Mat product of CV_32F matrices, float accumulate...
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0;                                    // (cv::gemm accumulates CV_32F products in double)
            for (int k = 0; k < 4; ++k) s += (double)bTc.v[4 * r + k] * (double)Tcw.v[4 * k + c];
            T[4 * r + c] = (float)s;
        }
    float RT[9], t[3];                                       // cvu::inv (cvutil.cpp:15-23): R^T and -R^T t
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) RT[3 * r + c] = T[4 * c + r];
    for (int r = 0; r < 3; ++r) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += (double)(-RT[3 * r + k]) * (double)T[4 * k + 3];
        t[r] = (float)s;
    }
    // Eigen::Quaterniond(Matrix3d) (Shepperd's branches as in Eigen/src/Geometry/Quaternion.h)
    double R[9];
    for (int k = 0; k < 9; ++k) R[k] = (double)RT[k];
    double q0, q1, q2, q3;
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0);
        q0 = 0.5 * s;
        s = 0.5 / s;
        q1 = (R[7] - R[5]) * s; q2 = (R[2] - R[6]) * s; q3 = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        double q[4];
        q[1 + i] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * s;
        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
    }
    TrajectoryEntry e;
    e.id = id;
    e.x = t[0]; e.y = t[1]; e.z = t[2];
    e.yaw = std::atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
    return e;
}
inline std::string trajectoryText(const StoredMap& map, const MatF& bTc) {
    std::string out;
    char buf[160];
    for (const StoredKeyFrame& kf : map.kfs) {
        if (kf.null) continue;
        const TrajectoryEntry e = trajectoryEntry(kf.id, bTc, kf.Tcw);
        std::snprintf(buf, sizeof buf, "%d %g %g %g %g\n", e.id, (double)e.x, (double)e.y, (double)e.z, e.yaw);
        out += buf;
    }
    return out;
}
inline void saveKeyFrameTrajectory(const std::string& path, const StoredMap& map, const MatF& bTc) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("cannot write " + path);
    const std::string t = trajectoryText(map, bTc);
    f.write(t.data(), (std::streamsize)t.size());
}
inline std::vector<TrajectoryEntry> loadKeyFrameTrajectory(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot read " + path);
    std::vector<TrajectoryEntry> out;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        TrajectoryEntry e;
        double x, y, z;
        if (std::sscanf(line.c_str(), "%d %lf %lf %lf %lf", &e.id, &x, &y, &z, &e.yaw) != 5)
            throw std::runtime_error("trajectory line '" + line + "' is not 'id x y z yaw'");
        e.x = (float)x; e.y = (float)y; e.z = (float)z;
        out.push_back(e);
    }
    return out;
}

}  // namespace se2lam_amd
