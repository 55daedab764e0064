// scenario.cpp -- Scenario::from_toml, the serde(Deserialize) behaviour of scenario.rs:9-66.
#include "pedoni_host.hpp"
#include "toml_lite.hpp"

#include <stdexcept>

namespace pedoni_host {
namespace {

using toml::Table;
using toml::Value;

[[noreturn]] void bad(const std::string& msg) { throw std::runtime_error("scenario: " + msg); }

const Value& require(const Table& t, const char* key)
{
    auto it = t.find(key);
    if (it == t.end()) bad(std::string("missing field `") + key + "`");
    return it->second;
}

float as_f32(const Value& v, const char* what)
{
    if (!v.is_number()) bad(std::string("`") + what + "` must be a number");
    return (float)v.as_double();
}

Vec2 as_vec2(const Value& v, const char* what)
{
    if (v.kind != Value::ArrayK || v.arr->size() != 2)
        bad(std::string("`") + what + "` must be [x, y]");
    Vec2 r;
    r.x = as_f32((*v.arr)[0], what);
    r.y = as_f32((*v.arr)[1], what);
    return r;
}

void as_line(const Value& v, Vec2 out[2])
{
    if (v.kind != Value::ArrayK || v.arr->size() != 2) bad("`line` must hold two points");
    out[0] = as_vec2((*v.arr)[0], "line");
    out[1] = as_vec2((*v.arr)[1], "line");
}

const Table& as_table(const Value& v, const char* what)
{
    if (v.kind != Value::TableK) bad(std::string("`") + what + "` must be a table");
    return *v.tab;
}

const toml::Array& as_table_array(const Value& v, const char* what)
{
    if (v.kind != Value::ArrayK) bad(std::string("`") + what + "` must be an array");
    return *v.arr;
}

size_t as_usize(const Value& v, const char* what)
{
    if (v.kind != Value::Integer || v.i < 0) bad(std::string("`") + what + "` must be a non-negative integer");
    return (size_t)v.i;
}

} // namespace

Scenario Scenario::from_toml(const std::string& text)
{
    Table doc;
    try {
        doc = toml::parse(text);
    } catch (const toml::ParseError& e) {
        bad(e.what());
    }
    Scenario sc;
    sc.field.size = as_vec2(require(as_table(require(doc, "field"), "field"), "size"), "size");

    for (const Value& w : as_table_array(require(doc, "waypoints"), "waypoints")) {
        const Table& t = as_table(w, "waypoints");
        WaypointConfig c;
        as_line(require(t, "line"), c.line);
        auto it = t.find("width");
        c.width = it == t.end() ? 1.0f : as_f32(it->second, "width"); // scenario.rs:3-5,41-42
        sc.waypoints.push_back(c);
    }
    for (const Value& o : as_table_array(require(doc, "obstacles"), "obstacles")) {
        const Table& t = as_table(o, "obstacles");
        ObstacleConfig c;
        as_line(require(t, "line"), c.line);
        auto it = t.find("width");
        c.width = it == t.end() ? 1.0f : as_f32(it->second, "width"); // scenario.rs:25-26
        sc.obstacles.push_back(c);
    }
    for (const Value& p : as_table_array(require(doc, "pedestrians"), "pedestrians")) {
        const Table& t = as_table(p, "pedestrians");
        PedestrianConfig c;
        c.origin = as_usize(require(t, "origin"), "origin");
        c.destination = as_usize(require(t, "destination"), "destination");
        const Table& s = as_table(require(t, "spawn"), "spawn");
        const Value& kind = require(s, "kind");
        if (kind.kind != Value::String) bad("`kind` must be a string");
        if (kind.s == "periodic") {                       // scenario.rs:63
            c.spawn.kind = PedestrianSpawnConfig::Periodic;
            const Value& f = require(s, "frequency");
            if (!f.is_number()) bad("`frequency` must be a number");
            c.spawn.frequency = f.as_double();
        } else if (kind.s == "once") {                    // scenario.rs:64
            c.spawn.kind = PedestrianSpawnConfig::Once;
            const Value& n = require(s, "count");
            if (n.kind != Value::Integer) bad("`count` must be an integer");
            c.spawn.count = (int32_t)n.i;
        } else {
            bad("unknown variant `" + kind.s + "`, expected `periodic` or `once`");
        }
        sc.pedestrians.push_back(c);
    }
    return sc;
}

} // namespace pedoni_host
