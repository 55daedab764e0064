// toml_lite.hpp -- the subset of TOML 1.0 that pedoni scenario files use.
//
// The reference parses scenarios with the `toml` crate (pedoni/src/main.rs:55).  Its
// scenario files (scenarios/*.toml) use: comments, [table] and [[array-of-tables]]
// headers, bare keys, integers, floats, basic strings, booleans, (nested, multi-line,
// trailing-comma) arrays and inline tables.  That is what this reader accepts; dotted
// keys, dates, multi-line / literal strings are rejected with an error, never guessed.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pedoni_host {
namespace toml {

struct Value;
using Table = std::map<std::string, Value>;
using Array = std::vector<Value>;

struct Value {
    enum Kind { Integer, Float, String, Bool, ArrayK, TableK } kind = Integer;
    int64_t i = 0;
    double f = 0.0;
    bool b = false;
    std::string s;
    std::shared_ptr<Array> arr;
    std::shared_ptr<Table> tab;
    bool array_of_tables = false; // created by [[header]]

    bool is_number() const { return kind == Integer || kind == Float; }
    // serde coerces TOML integers into f32/f64 fields (scenarios/narrow-gap.toml:2)
    double as_double() const { return kind == Integer ? (double)i : f; }
};

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

Table parse(const std::string& text);

} // namespace toml
} // namespace pedoni_host
