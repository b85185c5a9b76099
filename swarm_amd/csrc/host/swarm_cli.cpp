// swarm_cli.cpp -- the `swarm` command line tool over the MI355X kernels.
//
// Same surface as the reference's tool (swarm/swarm.cpp: the commands integrate, benchmark, verify, query, test,
// generate, convert; the options of L454-495; bare key=value words go into the configuration; the CSV columns and
// the notices are the ones its users and its ctest scripts parse), organised differently: an option table, a
// command table, a Session object that owns configuration and ensembles, a Sweep that expands --range into the
// values to run, and Deviation/Tolerance for every "is this run close enough" question.
#include <sys/time.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <sstream>

#include "swarm/swarm.h"

namespace {

using swarm::config;
using swarm::defaultEnsemble;
using std::string;

// ---- small pieces -----------------------------------------------------------------------------------------------
// wall-clock milliseconds of `work`, the device idle on both sides (the reference's stopwatch does the same, so
// "Integration (ms)" covers upload, every launch, log flushes, the active counts and the download)
double milliseconds(const std::function<void()> &work) {
    auto now = [] {
        swarm_hip_synchronize(0);
        timeval t;
        gettimeofday(&t, 0);
        return t.tv_sec * 1e3 + t.tv_usec * 1e-3;
    };
    const double t0 = now();
    work();
    return now() - t0;
}

struct Tolerance {
    double pos, vel, time;
    explicit Tolerance(const config &cfg) // defaults of swarm.cpp:572-574
        : pos(cfg.optional("pos_threshold", 1e-10)), vel(cfg.optional("vel_threshold", 1e-10)),
          time(cfg.optional("time_threshold", 1e-10)) {}
};

// distance between two ensembles in the three measures the tool reports
struct Deviation {
    double pos = 0, vel = 0, time = 0;
    bool comparable = true;
    Deviation() {}
    Deviation(defaultEnsemble &a, defaultEnsemble &b) { comparable = swarm::compare_ensembles(a, b, pos, vel, time); }
    bool within(const Tolerance &t) const { return comparable && !(pos > t.pos) && !(vel > t.vel) && !(time > t.time); }
};

// An ensemble file named by the configuration: the binary key wins over the text key.
struct EnsembleFile {
    string path;
    bool text = false;
    static bool named(const config &cfg, const char *binary_key, const char *text_key, EnsembleFile &f) {
        if (cfg.valid(binary_key)) return f.path = cfg.at(binary_key), f.text = false, true;
        if (cfg.valid(text_key)) return f.path = cfg.at(text_key), f.text = true, true;
        return false;
    }
    static bool input(const config &cfg, EnsembleFile &f) { return named(cfg, "input", "text_input", f); }
    static bool output(const config &cfg, EnsembleFile &f) { return named(cfg, "output", "text_output", f); }

    defaultEnsemble read() const {
        INFO_OUTPUT(1, "Loading from " << (text ? "text" : "binary") << " file " << path);
        defaultEnsemble e = text ? swarm::snapshot::load_text(path) : swarm::snapshot::load(path);
        INFO_OUTPUT(1, ", time = " << e.time_ranges() << std::endl);
        return e;
    }
    void write(defaultEnsemble &e) const {
        INFO_OUTPUT(1, "Saving as " << (text ? "text" : "binary") << "  " << path);
        INFO_OUTPUT(1, ", time = " << e.time_ranges() << std::endl);
        if (text)
            swarm::snapshot::save_text(e, path);
        else
            swarm::snapshot::save(e, path);
    }
};

// ---- --range param=a..b | a..b..step | v1,v2,... ------------------------------------------------------------------
// (the grammar of swarm.cpp:338-379: numbers are [0-9.Ee]+, list items [A-Za-z0-9_.]+, a list has at least two items)
class Sweep {
public:
    string parameter;
    std::vector<string> values; // every value the parameter takes, in order, as the text that goes into the cfg

    explicit Sweep(const string &spec) {
        const size_t eq = spec.find('=');
        if (eq == string::npos || eq == 0 || eq + 1 == spec.size() || !all_of(spec.substr(0, eq), is_word))
            ERROR("Wrong parameter-value pair ");
        parameter = spec.substr(0, eq);
        const string rhs = spec.substr(eq + 1);
        std::vector<string> parts = split(rhs, "..");
        if ((parts.size() == 2 || parts.size() == 3) && every(parts, is_number_char)) {
            const double first = atof(parts[0].c_str()), last = atof(parts[1].c_str());
            const double step = parts.size() == 3 ? atof(parts[2].c_str()) : 1;
            for (double v = first; v <= last; v += step) { // values as the stream prints them (6 significant digits)
                std::ostringstream text;
                text << v;
                values.push_back(text.str());
            }
            return;
        }
        parts = split(rhs, ",");
        if (parts.size() >= 2 && every(parts, is_item_char)) {
            values = parts;
            return;
        }
        ERROR("Range is invalid");
    }
    // does a change of this parameter need a new initial ensemble?
    bool reshapes_ensemble() const { return parameter == "input" || parameter == "nsys" || parameter == "nbod"; }

private:
    static bool is_word(char c) { return isalnum((unsigned char)c) || c == '_'; }
    static bool is_number_char(char c) { return isdigit((unsigned char)c) || c == '.' || c == 'E' || c == 'e'; }
    static bool is_item_char(char c) { return is_word(c) || c == '.'; }
    static bool all_of(const string &s, bool (*ok)(char)) {
        if (s.empty()) return false;
        for (char c : s)
            if (!ok(c)) return false;
        return true;
    }
    static bool every(const std::vector<string> &v, bool (*ok)(char)) {
        for (const string &s : v)
            if (!all_of(s, ok)) return false;
        return true;
    }
    static std::vector<string> split(const string &s, const string &sep) {
        std::vector<string> out;
        size_t at = 0;
        for (size_t hit; (hit = s.find(sep, at)) != string::npos; at = hit + sep.size()) out.push_back(s.substr(at, hit - at));
        out.push_back(s.substr(at));
        return out;
    }
};

// ---- one invocation of the tool -------------------------------------------------------------------------------------
class Session {
public:
    config cfg;
    std::map<string, string> options; // command-line options by long name
    string command;

    int run();

private:
    defaultEnsemble initial, current, reference;
    swarm::Pintegrator integ;

    void require_valid_configuration() {
        if (!swarm::validate_configuration(cfg)) ERROR("Invalid configuration");
    }
    // the starting ensemble: the named input file, else a generated one
    void obtain_initial(bool must_be_a_file = false) {
        EnsembleFile f;
        if (EnsembleFile::input(cfg, f)) {
            initial = f.read();
        } else if (must_be_a_file) {
            ERROR("you should have a tested input file");
        } else {
            INFO_OUTPUT(1, "Generating new ensemble:  " << cfg["nsys"] << ", " << cfg["nbod"] << std::endl);
            initial = swarm::generate_ensemble(cfg);
        }
    }
    bool store(defaultEnsemble &e) {
        EnsembleFile f;
        if (!EnsembleFile::output(cfg, f)) return false;
        f.write(e);
        return true;
    }
    // an integrator for `target`, aimed at destination_time (default: mean start time + 10 pi, swarm.cpp:182-183)
    void make_integrator(defaultEnsemble &target) {
        const double t_end = cfg.optional("destination_time", initial.time_ranges().average + 10 * M_PI);
        integ = swarm::integrator::create(cfg);
        integ->set_max_iterations(cfg.optional("max_iterations", swarm::integrator::_default_max_iterations));
        integ->set_ensemble(target);
        integ->set_destination_time(t_end);
    }
    // -n/--interval: integrate in stages and print the energy error after each; give up (exit 1) when the median
    // error passes allowed_deltaE.  Without -n: one integrate().
    void advance() {
        if (!cfg.valid("interval")) {
            integ->integrate();
            return;
        }
        const double t_begin = current.time_ranges().median;
        const double t_end = cfg.optional("destination_time", t_begin + 10 * M_PI);
        const double stage = cfg.optional("interval", t_end - t_begin);
        const double base = cfg.optional("logarithmic", 0.0);
        const double limit = cfg.optional("allowed_deltaE", 0.01);
        if (t_end < t_begin) ERROR("Destination time should be larger than begin time");
        if (stage < 0) ERROR("Interval cannot be negative");
        if (stage < 0.001) ERROR("Interval too small");
        if (base != 0 && base <= 1) ERROR("logarithm base should be greater than 1");
        std::cout << "Time, Energy Conservation Factor (delta E/E), Active Systems" << std::endl;
        double mark = t_begin;
        while (mark < t_end) {
            mark = base > 1 ? (mark > 0 ? mark * base : stage) : mark + stage;
            const double upto = std::min(mark, t_end);
            integ->set_destination_time(upto);
            integ->integrate();
            const swarm::ensemble::range_t err = swarm::energy_conservation_error_range(current, initial);
            std::cout << upto << ", " << err.max << ", " << err.median << ", "
                      << current.nsys() - swarm::number_of_disabled_systems(current) << std::endl;
            if (err.median > limit) {
                INFO_OUTPUT(0, "At least half of systems are too unstable to conserve energy. dE/E: " << err.median << std::endl);
                exit(1);
            }
        }
    }

    int cmd_integrate();
    int cmd_test();
    int cmd_sweep(bool verify);
    int cmd_query();
    int cmd_generate();
    int cmd_convert();
};

int Session::cmd_integrate() {
    require_valid_configuration();
    obtain_initial();
    current = initial.clone();
    make_integrator(current);
    const double ms = milliseconds([this] { advance(); });
    store(current);
    INFO_OUTPUT(1, "Integration time: " << ms << " ms " << std::endl);
    return 0;
}

// integrate the input file and hold the result against the expected-output file (what ctest runs for every plugin)
int Session::cmd_test() {
    require_valid_configuration();
    obtain_initial(true);
    current = initial.clone();
    make_integrator(current);
    const double ms = milliseconds([this] { advance(); });
    EnsembleFile expected;
    if (!EnsembleFile::output(cfg, expected)) ERROR("You should provide a test output file");
    reference = expected.read();
    const Deviation d(current, reference);
    INFO_OUTPUT(1, "\tPosition difference: " << d.pos << std::endl
                                              << "\tVelocity difference: " << d.vel << std::endl
                                              << "\tTime     difference: " << d.time << std::endl);
    if (!d.within(Tolerance(cfg))) {
        INFO_OUTPUT(0, "Test failed" << std::endl);
        exit(1);
    }
    INFO_OUTPUT(0, "Test success" << std::endl);
    INFO_OUTPUT(1, "Integration time: " << ms << " ms " << std::endl);
    return 0;
}

// benchmark / verify: one CSV row per value of the swept parameter.  verify first integrates the unswept
// configuration as the reference every row is compared with, and stops at the first row out of tolerance.
int Session::cmd_sweep(bool verify) {
    if (!options.count("range")) {
        std::cerr << "A parameter-value range is missing" << std::endl;
        if (verify) std::cout << "Verify success" << std::endl;
        return 0;
    }
    const Sweep sweep(options["range"]);
    const config unswept = cfg;
    const bool fresh_ensemble_per_row = sweep.reshapes_ensemble() || cfg.count("reinitialize");
    bool all_within = true;

    std::cout << "Parameter, Value, Time, Energy Conservation Factor (delta E/E)"
                 ", Position Difference, Velocity Difference, Time difference "
                 ", Integration (ms), Integrator initialize (ms) \n";
    if (!fresh_ensemble_per_row) obtain_initial();
    swarm::outputConfigSummary(std::cout, cfg);
    if (verify) {
        if (initial.empty()) obtain_initial();
        reference = initial.clone();
        make_integrator(reference);
        integ->integrate();
    }
    for (const string &value : sweep.values) {
        if (sweep.parameter == "config")
            cfg = config::load(value, unswept);
        else
            cfg[sweep.parameter] = value;
        require_valid_configuration();
        if (fresh_ensemble_per_row) obtain_initial();
        current = initial.clone();
        const double setup_ms = milliseconds([this] { make_integrator(current); });
        const double run_ms = milliseconds([this] { integ->integrate(); });
        const double energy_error = swarm::find_max_energy_conservation_error(current, initial);
        const Deviation d = reference.empty() ? Deviation() : Deviation(current, reference);
        std::cout << sweep.parameter << ", " << value << ",  " << current.time_ranges() << ",  " << energy_error << ",    "
                  << d.pos << ",    " << d.vel << ",   " << d.time << ",   " << run_ms << ",    " << setup_ms << std::endl;
        if (!d.within(Tolerance(unswept))) {
            all_within = false;
            if (verify) {
                INFO_OUTPUT(0, "Verify failed" << std::endl);
                exit(1);
            }
        }
    }
    if (verify) std::cout << (all_within ? "Verify success" : "Verify failed") << std::endl;
    return (verify && !all_within) ? 1 : 0;
}

int Session::cmd_query() {
    namespace q = swarm::query;
    if (!options.count("logfile")) {
        std::cerr << "Name of input log file is missing \n";
        return 1;
    }
    q::time_range_t times;
    q::sys_range_t systems;
    q::body_range_t bodies;
    if (options.count("time")) times = q::time_range_t::parse(options["time"]);
    if (options.count("system")) systems = q::sys_range_t::parse(options["system"]);
    if (options.count("body")) bodies = q::body_range_t::parse(options["body"]);
    // the last of these in the reference's order of tests wins
    const std::pair<const char *, q::planets_coordinate_system_t> frames[] = {
        {"origin", q::origin}, {"astrocentric", q::astrocentric}, {"barycentric", q::barycentric}, {"jacobi", q::jacobi}};
    for (const auto &f : frames)
        if (options.count(f.first)) q::set_coordinate_system(f.second);
    std::cout.flush();
    if (options.count("keplerian")) q::set_keplerian_output();
    std::cerr << "# Output in " << (options.count("keplerian") ? "Keplerian" : "Cartesian") << " coordinates  \n";
    q::execute(options["logfile"], times, systems, bodies);
    return 0;
}

int Session::cmd_generate() {
    INFO_OUTPUT(1, "Generating new ensemble:  " << cfg["nsys"] << ", " << cfg["nbod"] << std::endl);
    current = swarm::generate_ensemble(cfg);
    store(current);
    return 0;
}

int Session::cmd_convert() {
    EnsembleFile in;
    if (EnsembleFile::input(cfg, in)) {
        current = in.read();
        if (store(current)) {
            INFO_OUTPUT(1, "Converted!");
            return 0;
        }
    }
    INFO_OUTPUT(1, "Convertion failed");
    return 1;
}

int Session::run() {
    struct Entry {
        const char *name;
        bool uses_device;
        std::function<int()> go;
    };
    const Entry table[] = {{"integrate", true, [this] { return cmd_integrate(); }},
                           {"test", true, [this] { return cmd_test(); }},
                           {"benchmark", true, [this] { return cmd_sweep(false); }},
                           {"verify", true, [this] { return cmd_sweep(true); }},
                           {"query", false, [this] { return cmd_query(); }},
                           {"generate", false, [this] { return cmd_generate(); }},
                           {"convert", false, [this] { return cmd_convert(); }}};
    for (const Entry &e : table) {
        if (command != e.name) continue;
        if (e.uses_device) {
            // nogpu=1 selects the reference's CPU plugins (swarm.h:49-60).  This library has none: the key is accepted,
            // so that the reference's CPU configurations run unchanged, and every integrator plugin -- the *_cpu
            // names included -- runs on the MI355X.
            if (cfg.optional("nogpu", 0) && DEBUG_LEVEL >= 0)
                std::cerr << "# nogpu=1 ignored: no CPU integrators in this build, every integrator plugin runs on the MI355X"
                          << std::endl;
            swarm::init(cfg);
        }
        return e.go();
    }
    std::cerr << "Valid commands are: integrate, benchmark, verify, test, query, generate " << std::endl;
    return 0;
}

// ---- command line ---------------------------------------------------------------------------------------------------
const char *USAGE =
    "Usage:\n \tswarm [options] COMMAND [PARAMETER VALUE VALUE ...]\n\n"
    "Possible commands are: \n"
    "\tintegrate :  Integrate a [loaded|generated] ensemble\n"
    "\tbenchmark :  Compare outputs for different methods of integrations\n"
    "\tverify    :  Verify an integrator against a reference integrator\n"
    "\tquery     :  Query data from a log file\n"
    "\ttest      :  Test a configuration against input/output files\n"
    "\tgenerate  :  Generate a new ensemble and save it to output file\n"
    "\tconvert   :  Read input file and write it to output file (converts to/from text)\n"
    "\nOptions\n"
    "General Options:\n"
    "  -c [ --cfg ] arg              Integrator configuration file\n"
    "  --defaults                    Use default values for required configuration options\n"
    "  -h [ --help ]                 produce help message\n"
    "  -p [ --plugins ]              list all of the plugins\n"
    "  -v [ --verbose ] arg          Verbosity level (debug output) \n"
    "  -q [ --quiet ]                Suppress all the notifications (for CSV generation)\n"
    "  -V [ --version ]              Print version message\n"
    "Integation Options:\n"
    "  -d [ --destination_time ] arg Destination time to achieve\n"
    "  -l [ --logarithmic ] arg      Produce times in logarithmic scale\n"
    "  -n [ --interval ] arg         Energy test intervals\n"
    "  -i [ --input ] arg            Input file\n"
    "  -o [ --output ] arg           Output file\n"
    "  -I [ --text_input ] arg       Text Input file\n"
    "  -O [ --text_output ] arg      Text Output file\n"
    "Benchmark Options:\n"
    "  --range arg                   Parameter value range param=v1,v2,v3\n"
    "Query Options:\n"
    "  -t [ --time ] arg             range of times to query\n"
    "  -s [ --system ] arg           range of systems to query\n"
    "  -b [ --body ] arg             range of bodies to query\n"
    "  -k [ --keplerian ]            output in Keplerian coordinates\n"
    "  --astrocentric                output coordinates in astrocentric frame\n"
    "  --barycentric                 output coordinates in barycentric frame\n"
    "  --origin                      output coordinates in origin frame\n"
    "  --jacobi                      output coordinates in Jacobi frame [default]\n"
    "  -f [ --logfile ] arg          the log file to query\n";

struct OptionSpec {
    const char *name;
    char letter;       // 0: long form only
    bool takes_value;
    bool into_cfg;     // copied into the configuration under its own name (swarm.cpp:536-540)
};
const OptionSpec OPTIONS[] = {
    {"destination_time", 'd', true, true}, {"logarithmic", 'l', true, true}, {"interval", 'n', true, true},
    {"input", 'i', true, true},            {"output", 'o', true, true},      {"text_input", 'I', true, true},
    {"text_output", 'O', true, true},      {"range", 0, true, false},        {"cfg", 'c', true, false},
    {"nogpu", 'g', false, false},          {"defaults", 0, false, false},    {"help", 'h', false, false},
    {"plugins", 'p', false, false},        {"verbose", 'v', true, false},    {"quiet", 'q', false, false},
    {"version", 'V', false, false},        {"time", 't', true, false},       {"system", 's', true, false},
    {"body", 'b', true, false},            {"logfile", 'f', true, false},    {"astrocentric", 0, false, false},
    {"barycentric", 0, false, false},      {"origin", 0, false, false},      {"jacobi", 0, false, false},
    {"keplerian", 'k', false, false}};

const OptionSpec *find_option(const string &word, string &inline_value, bool &has_inline_value) {
    has_inline_value = false;
    if (word.size() > 2 && word.compare(0, 2, "--") == 0) {
        string name = word.substr(2);
        const size_t eq = name.find('=');
        if (eq != string::npos) inline_value = name.substr(eq + 1), name.erase(eq), has_inline_value = true;
        for (const OptionSpec &o : OPTIONS)
            if (name == o.name) return &o;
        ERROR("unrecognised option '" + word + "'");
    }
    for (const OptionSpec &o : OPTIONS)
        if (o.letter && word[1] == o.letter) {
            if (word.size() > 2) inline_value = word.substr(2), has_inline_value = true;
            return &o;
        }
    ERROR("unrecognised option '" + word + "'");
}

// fills `s` from argv; returns false when the invocation was answered here (help, version, plugin list)
bool read_command_line(int argc, char *argv[], Session &s, int &exit_code) {
    std::vector<string> words;
    for (int i = 1; i < argc; i++) {
        const string w = argv[i];
        // "-5", "-.5": a negative number is a word, not an option
        const bool option = w.size() >= 2 && w[0] == '-' && (w[1] == '-' ? w.size() > 2 : (!isdigit((unsigned char)w[1]) && w[1] != '.'));
        if (!option) {
            words.push_back(w);
            continue;
        }
        string value;
        bool have = false;
        const OptionSpec *o = find_option(w, value, have);
        if (!o->takes_value) {
            s.options[o->name] = "1";
            continue;
        }
        if (!have) {
            if (i + 1 >= argc) ERROR(string("the required argument for option '--") + o->name + "' is missing");
            value = argv[++i];
        }
        s.options[o->name] = value;
    }
    auto given = [&](const char *k) { return s.options.count(k) != 0; };
    exit_code = 0;
    if (given("help")) return std::cout << USAGE << "\n", exit_code = 1, false;
    if (given("version"))
        return std::cout << "Swarm is running as " << (sizeof(void *) == 8 ? "64bit" : "32bit") << std::endl, false;
    if (given("verbose")) DEBUG_LEVEL = atoi(s.options["verbose"].c_str());
    if (given("quiet")) DEBUG_LEVEL = -1;
    if (given("plugins")) return std::cout << swarm::plugin::help, false;
    if (given("defaults")) s.cfg = swarm::default_config();
    if (given("cfg")) s.cfg = config::load(s.options["cfg"], s.cfg);
    if (words.empty()) return std::cout << USAGE << "\n", exit_code = 1, false;
    s.command = words[0];
    for (const OptionSpec &o : OPTIONS)
        if (o.into_cfg && given(o.name)) s.cfg[o.name] = s.options[o.name];
    if (given("nogpu")) s.cfg["nogpu"] = "1";
    for (size_t i = 1; i < words.size(); i++) {
        const size_t eq = words[i].find('=');
        if (eq == string::npos) ERROR(("Wrong parameter value pair : " + words[i]).c_str());
        s.cfg[words[i].substr(0, eq)] = words[i].substr(eq + 1);
    }
    return true;
}

} // namespace

int main(int argc, char *argv[]) {
    try {
        Session session;
        int code = 0;
        if (!read_command_line(argc, argv, session, code)) return code;
        return session.run();
    } catch (std::exception &e) { // what an uncaught swarm::runtime_error prints in the reference's tool
        std::cerr << "terminate called after throwing an instance of 'swarm::runtime_error'\n  what():  " << e.what() << std::endl;
        return 1;
    }
}
