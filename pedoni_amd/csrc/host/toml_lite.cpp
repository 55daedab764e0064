#include "toml_lite.hpp"

#include <cctype>
#include <cstdlib>

namespace pedoni_host {
namespace toml {
namespace {

struct Parser {
    const std::string& src;
    size_t pos = 0;
    int line = 1;

    explicit Parser(const std::string& s) : src(s) {}

    [[noreturn]] void fail(const std::string& msg) const
    {
        throw ParseError("TOML parse error at line " + std::to_string(line) + ": " + msg);
    }
    bool eof() const { return pos >= src.size(); }
    char peek() const { return eof() ? '\0' : src[pos]; }
    char get()
    {
        char c = peek();
        if (c == '\n') ++line;
        ++pos;
        return c;
    }
    void skip_inline_ws()
    {
        while (peek() == ' ' || peek() == '\t') ++pos;
    }
    void skip_comment()
    {
        if (peek() == '#')
            while (!eof() && peek() != '\n') ++pos;
    }
    // whitespace, newlines and comments (allowed between array elements)
    void skip_ws_nl()
    {
        for (;;) {
            char c = peek();
            if (c == ' ' || c == '\t' || c == '\r' || c == '\n') get();
            else if (c == '#') skip_comment();
            else break;
        }
    }
    void expect_line_end()
    {
        skip_inline_ws();
        skip_comment();
        if (peek() == '\r') ++pos;
        if (eof()) return;
        if (peek() != '\n') fail(std::string("unexpected '") + peek() + "' after value");
        get();
    }

    std::string parse_key()
    {
        skip_inline_ws();
        std::string k;
        if (peek() == '"') {
            k = parse_string();
        } else {
            while (std::isalnum((unsigned char)peek()) || peek() == '_' || peek() == '-') k += get();
        }
        if (k.empty()) fail("expected a key");
        skip_inline_ws();
        if (peek() == '.') fail("dotted keys are not supported");
        return k;
    }

    std::string parse_string()
    {
        if (get() != '"') fail("expected '\"'");
        if (peek() == '"' && pos + 1 < src.size() && src[pos + 1] == '"')
            fail("multi-line strings are not supported");
        std::string out;
        for (;;) {
            if (eof() || peek() == '\n') fail("unterminated string");
            char c = get();
            if (c == '"') break;
            if (c == '\\') {
                char e = get();
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                default: fail("unsupported escape sequence");
                }
            } else {
                out += c;
            }
        }
        return out;
    }

    Value parse_number()
    {
        size_t start = pos;
        bool is_float = false;
        if (peek() == '+' || peek() == '-') ++pos;
        if (src.compare(pos, 3, "inf") == 0 || src.compare(pos, 3, "nan") == 0) {
            pos += 3;
            is_float = true;
        } else {
            if (!std::isdigit((unsigned char)peek())) fail("expected a number");
            while (std::isdigit((unsigned char)peek()) || peek() == '_') ++pos;
            if (peek() == '.') {
                is_float = true;
                ++pos;
                if (!std::isdigit((unsigned char)peek())) fail("digit expected after '.'");
                while (std::isdigit((unsigned char)peek()) || peek() == '_') ++pos;
            }
            if (peek() == 'e' || peek() == 'E') {
                is_float = true;
                ++pos;
                if (peek() == '+' || peek() == '-') ++pos;
                if (!std::isdigit((unsigned char)peek())) fail("digit expected in exponent");
                while (std::isdigit((unsigned char)peek())) ++pos;
            }
        }
        std::string tok;
        for (size_t k = start; k < pos; ++k)
            if (src[k] != '_') tok += src[k];
        Value v;
        if (is_float) {
            v.kind = Value::Float;
            v.f = std::strtod(tok.c_str(), nullptr);
        } else {
            v.kind = Value::Integer;
            v.i = std::strtoll(tok.c_str(), nullptr, 10);
        }
        return v;
    }

    Value parse_array()
    {
        get(); // '['
        Value v;
        v.kind = Value::ArrayK;
        v.arr = std::make_shared<Array>();
        for (;;) {
            skip_ws_nl();
            if (peek() == ']') { get(); break; }
            v.arr->push_back(parse_value());
            skip_ws_nl();
            if (peek() == ',') { get(); continue; }
            if (peek() == ']') { get(); break; }
            fail("expected ',' or ']' in array");
        }
        return v;
    }

    Value parse_inline_table()
    {
        get(); // '{'
        Value v;
        v.kind = Value::TableK;
        v.tab = std::make_shared<Table>();
        skip_inline_ws();
        if (peek() == '}') { get(); return v; }
        for (;;) {
            std::string k = parse_key();
            if (get() != '=') fail("expected '=' in inline table");
            skip_inline_ws();
            if (v.tab->count(k)) fail("duplicate key `" + k + "`");
            (*v.tab)[k] = parse_value();
            skip_inline_ws();
            if (peek() == ',') { get(); continue; }
            if (peek() == '}') { get(); break; }
            fail("expected ',' or '}' in inline table");
        }
        return v;
    }

    Value parse_value()
    {
        skip_inline_ws();
        char c = peek();
        if (c == '"') {
            Value v;
            v.kind = Value::String;
            v.s = parse_string();
            return v;
        }
        if (c == '\'') fail("literal strings are not supported");
        if (c == '[') return parse_array();
        if (c == '{') return parse_inline_table();
        if (src.compare(pos, 4, "true") == 0) { pos += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
        if (src.compare(pos, 5, "false") == 0) { pos += 5; Value v; v.kind = Value::Bool; v.b = false; return v; }
        if (c == '+' || c == '-' || std::isdigit((unsigned char)c) || c == 'i' || c == 'n')
            return parse_number();
        fail(std::string("unexpected character '") + c + "'");
    }

    Table parse_document()
    {
        Table root;
        Table* current = &root;
        for (;;) {
            skip_ws_nl();
            if (eof()) break;
            if (peek() == '[') {
                get();
                bool aot = false;
                if (peek() == '[') { get(); aot = true; }
                std::string name = parse_key();
                if (get() != ']') fail("expected ']'");
                if (aot && get() != ']') fail("expected ']]'");
                expect_line_end();
                if (aot) {
                    auto it = root.find(name);
                    if (it == root.end()) {
                        Value a;
                        a.kind = Value::ArrayK;
                        a.arr = std::make_shared<Array>();
                        a.array_of_tables = true;
                        it = root.emplace(name, a).first;
                    } else if (!it->second.array_of_tables) {
                        fail("`" + name + "` is not an array of tables");
                    }
                    Value& slot = it->second;
                    Value t;
                    t.kind = Value::TableK;
                    t.tab = std::make_shared<Table>();
                    slot.arr->push_back(t);
                    current = slot.arr->back().tab.get();
                } else {
                    if (root.count(name)) fail("table `" + name + "` defined twice");
                    Value t;
                    t.kind = Value::TableK;
                    t.tab = std::make_shared<Table>();
                    root[name] = t;
                    current = root[name].tab.get();
                }
            } else {
                std::string k = parse_key();
                if (get() != '=') fail("expected '=' after key `" + k + "`");
                if (current->count(k)) fail("duplicate key `" + k + "`");
                Value v = parse_value();
                (*current)[k] = v;
                expect_line_end();
            }
        }
        return root;
    }
};

} // namespace

Table parse(const std::string& text)
{
    Parser p(text);
    return p.parse_document();
}

} // namespace toml
} // namespace pedoni_host
