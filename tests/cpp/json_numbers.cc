// The number grammar of include/MeshFEMHip/Json.hh: what nlohmann::json -- the reader of the reference's .bc / .material
// files -- rejects must be rejected here too (nan, inf, hex floats, a leading '+' or '.'), whatever the process's LC_NUMERIC.
#include <MeshFEMHip/Json.hh>
#include <cstdio>
#include <clocale>
using namespace MeshFEMHip;
int main() {
    setlocale(LC_NUMERIC, "de_DE.UTF-8");   // (a comma-decimal locale if the box has one: the conversion must not care)
    const char *good[] = {"1.5", "-0.25e+2", "0", "[1, 2.0E-3]", "{\"a\": -12.75}"};
    const char *bad[] = {"nan", "inf", "+1", ".5", "0x10", "1.", "1e", "01", "-", "1e400", "[1.5,]"};
    for (auto g : good) { try { Json j = Json::parse(g); printf("ok %s\n", g); } catch (const std::exception &e) { printf("REJECTED %s: %s\n", g, e.what()); } }
    for (auto b : bad) { try { Json j = Json::parse(b); printf("ACCEPTED %s\n", b); } catch (const std::exception &e) { printf("rejected %s\n", b); } }
    Json j = Json::parse("[1.5, -0.25e+2]");
    printf("%g %g\n", j[0].get<double>(), j[1].get<double>());
}
