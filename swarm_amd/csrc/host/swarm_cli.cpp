// swarm_cli.cpp -- the `swarm` command line tool over the MI355X kernels: integrate, test,
// benchmark, verify, generate, convert.  Mirrors swarm/swarm.cpp of the reference (same options,
// same key=value positional pairs, same CSV and messages); boost::program_options is replaced by
// a small hand-written parser.  `query` reads binary log files (no Berkeley DB, no -k output).
#include <sys/time.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <regex>
#include <sstream>

#include "swarm/swarm.h"

using namespace swarm;
using namespace std;

static string command;
static config cfg, base_cfg;
static defaultEnsemble initial_ens, current_ens, reference_ens;
static Pintegrator integ;
static map<string, string> argvars;

// stopwatch.h:134-142: wall clock around the callable, device synchronised
template <class F> static double watch_time(F f) {
    timeval a, b;
    swarm_hip_synchronize(0);
    gettimeofday(&a, 0);
    f();
    swarm_hip_synchronize(0);
    gettimeofday(&b, 0);
    return (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_usec - a.tv_usec) * 1e-3;
}

static void stability_test() { // swarm.cpp:61-105
    defaultEnsemble &ens = current_ens;
    double begin_time = ens.time_ranges().median;
    double destination_time = cfg.optional("destination_time", begin_time + 10 * M_PI);
    double interval = cfg.optional("interval", (destination_time - begin_time));
    double logarithmic = cfg.optional("logarithmic", 0.0);
    double allowed_deltaE = cfg.optional("allowed_deltaE", 0.01);
    if (destination_time < begin_time) ERROR("Destination time should be larger than begin time");
    if (interval < 0) ERROR("Interval cannot be negative");
    if (interval < 0.001) ERROR("Interval too small");
    if (logarithmic != 0 && logarithmic <= 1) ERROR("logarithm base should be greater than 1");
    std::cout << "Time, Energy Conservation Factor (delta E/E), Active Systems" << std::endl;
    for (double time = begin_time; time < destination_time;) {
        if (logarithmic > 1)
            time = (time > 0) ? time * logarithmic : interval;
        else
            time += interval;
        double effective_time = min(time, destination_time);
        integ->set_destination_time(effective_time);
        integ->integrate();
        ensemble::range_t deltaE_range = energy_conservation_error_range(ens, initial_ens);
        int active_systems = ens.nsys() - number_of_disabled_systems(ens);
        std::cout << effective_time << ", " << deltaE_range.max << ", " << deltaE_range.median << ", " << active_systems
                  << std::endl;
        if (deltaE_range.median > allowed_deltaE) {
            INFO_OUTPUT(0, "At least half of systems are too unstable to conserve energy. dE/E: " << deltaE_range.median
                                                                                                   << std::endl);
            exit(1);
        }
    }
}

static bool read_file(defaultEnsemble &ens, const char *bin_key, const char *text_key) { // swarm.cpp:107-143
    if (cfg.valid(bin_key)) {
        INFO_OUTPUT(1, "Loading from binary file " << cfg[bin_key]);
        ens = snapshot::load(cfg[bin_key]);
        INFO_OUTPUT(1, ", time = " << ens.time_ranges() << endl);
        return true;
    } else if (cfg.valid(text_key)) {
        INFO_OUTPUT(1, "Loading from text file " << cfg[text_key]);
        ens = snapshot::load_text(cfg[text_key]);
        INFO_OUTPUT(1, ", time = " << ens.time_ranges() << endl);
        return true;
    }
    return false;
}
static bool read_input_file(defaultEnsemble &ens) { return read_file(ens, "input", "text_input"); }
static bool read_output_file(defaultEnsemble &ens) { return read_file(ens, "output", "text_output"); }

static void load_generate_ensemble() { // swarm.cpp:145-154
    if (read_input_file(initial_ens)) {
    } else {
        INFO_OUTPUT(1, "Generating new ensemble:  " << cfg["nsys"] << ", " << cfg["nbod"] << endl);
        initial_ens = generate_ensemble(cfg);
    }
}

static bool save_ensemble() { // swarm.cpp:156-172
    if (cfg.valid("output")) {
        INFO_OUTPUT(1, "Saving as binary  " << cfg["output"]);
        INFO_OUTPUT(1, ", time = " << current_ens.time_ranges() << endl);
        snapshot::save(current_ens, cfg["output"]);
        return true;
    } else if (cfg.valid("text_output")) {
        INFO_OUTPUT(1, "Saving as text  " << cfg["text_output"]);
        INFO_OUTPUT(1, ", time = " << current_ens.time_ranges() << endl);
        snapshot::save_text(current_ens, cfg["text_output"]);
        return true;
    }
    return false;
}

static void prepare_integrator() { // swarm.cpp:179-190
    double begin_time = initial_ens.time_ranges().average;
    double destination_time = cfg.optional("destination_time", begin_time + 10 * M_PI);
    int max_iterations = cfg.optional("max_iterations", integrator::_default_max_iterations);
    integ = integrator::create(cfg);
    integ->set_max_iterations(max_iterations);
    integ->set_ensemble(current_ens);
    integ->set_destination_time(destination_time);
}
static void generic_integrate() { integ->integrate(); }
static void run_or_stability() {
    if (cfg.valid("interval"))
        stability_test();
    else
        generic_integrate();
}

static void run_integration() { // swarm.cpp:197-212
    if (!validate_configuration(cfg)) ERROR("Invalid configuration");
    load_generate_ensemble();
    current_ens = initial_ens.clone();
    prepare_integrator();
    double integration_time = watch_time(run_or_stability);
    save_ensemble();
    INFO_OUTPUT(1, "Integration time: " << integration_time << " ms " << std::endl);
}

static void reference_integration() { // swarm.cpp:215-222
    reference_ens = initial_ens.clone();
    prepare_integrator();
    integ->set_ensemble(reference_ens);
    integ->integrate();
}

static bool verification_results = true, verify_mode = false;
static double pos_threshold = 0, vel_threshold = 0, time_threshold = 0;

static void fail_verify() {
    if (verify_mode) {
        INFO_OUTPUT(0, "Verify failed" << endl);
        exit(1);
    } else
        verification_results = false;
}

static void output_test() { // swarm.cpp:238-275
    if (!validate_configuration(cfg)) ERROR("Invalid configuration");
    if (!read_input_file(initial_ens)) ERROR("you should have a tested input file");
    current_ens = initial_ens.clone();
    prepare_integrator();
    double integration_time = watch_time(run_or_stability);
    if (read_output_file(reference_ens)) {
        double pos_diff = 0, vel_diff = 0, time_diff = 0;
        bool comparison = compare_ensembles(current_ens, reference_ens, pos_diff, vel_diff, time_diff);
        INFO_OUTPUT(1, "\tPosition difference: " << pos_diff << endl
                                                  << "\tVelocity difference: " << vel_diff << endl
                                                  << "\tTime     difference: " << time_diff << endl);
        if (!comparison || pos_diff > pos_threshold || vel_diff > vel_threshold || time_diff > time_threshold) {
            INFO_OUTPUT(0, "Test failed" << endl);
            exit(1);
        } else {
            INFO_OUTPUT(0, "Test success" << endl);
        }
    } else {
        ERROR("You should provide a test output file");
    }
    INFO_OUTPUT(1, "Integration time: " << integration_time << " ms " << std::endl);
}

static void benchmark_item(const string &param, const string &value) { // swarm.cpp:277-313
    if (!validate_configuration(cfg)) ERROR("Invalid configuration");
    if (param == "input" || param == "nsys" || param == "nbod" || cfg.count("reinitialize")) load_generate_ensemble();
    current_ens = initial_ens.clone();
    double init_time = watch_time(prepare_integrator);
    double integration_time = watch_time(generic_integrate);
    double max_deltaE = find_max_energy_conservation_error(current_ens, initial_ens);
    double pos_diff = 0, vel_diff = 0, time_diff = 0;
    bool comparison = reference_ens.empty() ? true : compare_ensembles(current_ens, reference_ens, pos_diff, vel_diff, time_diff);
    std::cout << param << ", " << value << ",  " << current_ens.time_ranges() << ",  " << max_deltaE << ",    " << pos_diff
              << ",    " << vel_diff << ",   " << time_diff << ",   " << integration_time << ",    " << init_time << std::endl;
    if (!comparison || pos_diff > pos_threshold || vel_diff > vel_threshold || time_diff > time_threshold) fail_verify();
}

struct parameter_range {
    string parameter;
    vector<string> values;
    bool interpolate;
};

// the --range grammar, swarm.cpp:338-379
static parameter_range parse_range(const string &s) {
    parameter_range p;
    static regex assign("^(\\w+)=(.+)$");
    static regex range("^([0-9\\.Ee]+)\\.\\.([0-9\\.Ee]+)$");
    static regex rangeinc("^([0-9\\.Ee]+)\\.\\.([0-9\\.Ee]+)\\.\\.([0-9\\.Ee]+)$");
    static regex list("^([a-zA-Z0-9_\\.]+)(?:,([a-zA-Z0-9_\\.]+))+$");
    static regex item("([a-zA-Z0-9_\\.]+)");
    smatch match;
    if (regex_match(s, match, assign)) {
        p.parameter = match[1];
        string value_range = match[2];
        smatch m;
        if (regex_match(value_range, m, rangeinc)) {
            p.values.push_back(m[1]);
            p.values.push_back(m[3]);
            p.values.push_back(m[2]);
            p.interpolate = true;
        } else if (regex_match(value_range, m, range)) {
            p.values.push_back(m[1]);
            p.values.push_back(m[2]);
            p.interpolate = true;
        } else if (regex_match(value_range, m, list)) {
            for (sregex_iterator it(value_range.begin(), value_range.end(), item), end; it != end; ++it)
                p.values.push_back((*it)[0]);
            p.interpolate = false;
        } else
            ERROR("Range is invalid");
    } else
        ERROR("Wrong parameter-value pair ");
    return p;
}

static void benchmark(const parameter_range &pr) { // swarm.cpp:381-427
    std::cout << "Parameter, Value, Time, Energy Conservation Factor (delta E/E)"
                 ", Position Difference, Velocity Difference, Time difference "
                 ", Integration (ms), Integrator initialize (ms) \n";
    string param = pr.parameter;
    vector<string> values = pr.values;
    if (!(param == "input" || param == "nsys" || param == "nbod" || cfg.count("reinitialize"))) load_generate_ensemble();
    outputConfigSummary(std::cout, cfg);
    if (verify_mode) reference_integration();
    if (pr.interpolate) {
        double from = atof(values[0].c_str()), to = atof(values[1].c_str());
        double increment = (values.size() == 3) ? atof(values[2].c_str()) : 1;
        for (double i = from; i <= to; i += increment) {
            std::ostringstream stream;
            stream << i;
            string value = stream.str();
            cfg[param] = value;
            benchmark_item(param, value);
        }
    } else {
        for (vector<string>::const_iterator i = values.begin(); i != values.end(); i++) {
            string value = *i;
            if (param == "config")
                cfg = config::load(value, base_cfg);
            else
                cfg[param] = *i;
            benchmark_item(param, value);
        }
    }
}

static const char *USAGE =
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
    "  --astrocentric                output coordinates in astrocentric frame\n"
    "  --barycentric                 output coordinates in barycentric frame\n"
    "  --origin                      output coordinates in origin frame\n"
    "  --jacobi                      output coordinates in Jacobi frame [default]\n"
    "  -f [ --logfile ] arg          the log file to query\n";

static void parse_commandline_and_config(int argc, char *argv[]) { // swarm.cpp:435-559
    struct opt { const char *lng; char sht; bool arg; };
    static const opt opts[] = {{"destination_time", 'd', true}, {"logarithmic", 'l', true}, {"interval", 'n', true},
                               {"input", 'i', true}, {"output", 'o', true}, {"text_input", 'I', true},
                               {"text_output", 'O', true}, {"range", 0, true}, {"cfg", 'c', true}, {"nogpu", 'g', false},
                               {"defaults", 0, false}, {"help", 'h', false}, {"plugins", 'p', false},
                               {"verbose", 'v', true}, {"quiet", 'q', false}, {"version", 'V', false},
                               {"time", 't', true}, {"system", 's', true}, {"body", 'b', true}, {"logfile", 'f', true},
                               {"astrocentric", 0, false}, {"barycentric", 0, false}, {"origin", 0, false},
                               {"jacobi", 0, false}, {"keplerian", 'k', false}};
    vector<string> positional;
    for (int i = 1; i < argc; i++) {
        string a = argv[i];
        const opt *o = 0;
        string val;
        bool has_val = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            string name = a.substr(2);
            size_t eq = name.find('=');
            if (eq != string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
            for (const opt &c : opts) if (name == c.lng) o = &c;
            if (!o) ERROR("unrecognised option '" + a + "'");
        } else if (a.size() >= 2 && a[0] == '-' && !isdigit(a[1]) && a[1] != '.') {
            for (const opt &c : opts) if (c.sht && a[1] == c.sht) o = &c;
            if (!o) ERROR("unrecognised option '" + a + "'");
            if (a.size() > 2) { val = a.substr(2); has_val = true; }
        } else {
            positional.push_back(a);
            continue;
        }
        if (o->arg) {
            if (!has_val) {
                if (i + 1 >= argc) ERROR(string("the required argument for option '--") + o->lng + "' is missing");
                val = argv[++i];
            }
            argvars[o->lng] = val;
        } else
            argvars[o->lng] = "1";
    }
    if (argvars.count("help")) { std::cout << USAGE << "\n"; exit(1); }
    if (argvars.count("version")) { cout << "Swarm is running as " << (sizeof(void *) == 8 ? "64bit" : "32bit") << endl; exit(0); }
    if (argvars.count("verbose")) DEBUG_LEVEL = atoi(argvars["verbose"].c_str());
    if (argvars.count("quiet")) DEBUG_LEVEL = -1;
    if (argvars.count("plugins")) { cout << swarm::plugin::help; exit(0); }
    if (argvars.count("defaults")) cfg = default_config();
    if (argvars.count("cfg")) cfg = config::load(argvars["cfg"], cfg);
    if (positional.empty()) { std::cout << USAGE << "\n"; exit(1); }
    argvars["command"] = positional[0];
    const char *cmd_to_config[] = {"input", "output", "text_input", "text_output", "destination_time", "logarithmic", "interval"};
    for (const char *k : cmd_to_config) if (argvars.count(k)) cfg[k] = argvars[k];
    if (argvars.count("nogpu")) cfg["nogpu"] = "1";
    for (size_t i = 1; i < positional.size(); i++) {
        string::size_type pos = positional[i].find_first_of("=");
        if (string::npos != pos) {
            cfg[positional[i].substr(0, pos)] = positional[i].substr(pos + 1);
        } else
            ERROR(("Wrong parameter value pair : " + positional[i]).c_str());
    }
}

static int real_main(int argc, char *argv[]) {
    parse_commandline_and_config(argc, argv);
    base_cfg = cfg;
    command = argvars["command"];
    pos_threshold = cfg.optional("pos_threshold", 1e-10); // swarm.cpp:572-574
    vel_threshold = cfg.optional("vel_threshold", 1e-10);
    time_threshold = cfg.optional("time_threshold", 1e-10);
    // nogpu=1 selects the reference's CPU plugins (swarm.cpp:454-495, swarm.h:49-60).  This library has
    // none: the key is accepted, so that the reference's CPU configurations run unchanged, and the
    // integrator plugins -- the *_cpu names included -- run on the MI355X.
    if (cfg.optional("nogpu", 0) && command != "generate" && command != "convert" && command != "query" && DEBUG_LEVEL >= 0)
        cerr << "# nogpu=1 ignored: no CPU integrators in this build, every integrator plugin runs on the MI355X" << endl;

    if (command == "integrate") {
        swarm::init(cfg);
        run_integration();
    } else if (command == "test") {
        swarm::init(cfg);
        output_test();
    } else if (command == "benchmark" || command == "verify") {
        swarm::init(cfg);
        if (command == "verify") verify_mode = true;
        if (argvars.count("range"))
            benchmark(parse_range(argvars["range"]));
        else
            cerr << "A parameter-value range is missing" << endl;
        if (command == "verify") {
            cout << (verification_results ? "Verify success" : "Verify failed") << endl;
            return verification_results ? 0 : 1;
        }
    } else if (command == "query") { // swarm.cpp:602-646
        using namespace query;
        if (!argvars.count("logfile")) { cerr << "Name of input log file is missing \n"; return 1; }
        time_range_t T;
        sys_range_t sys, body_range;
        if (argvars.count("time")) T = time_range_t::parse(argvars["time"]);
        if (argvars.count("system")) sys = sys_range_t::parse(argvars["system"]);
        if (argvars.count("body")) body_range = body_range_t::parse(argvars["body"]);
        if (argvars.count("origin")) query::set_coordinate_system(query::origin);
        if (argvars.count("astrocentric")) query::set_coordinate_system(query::astrocentric);
        if (argvars.count("barycentric")) query::set_coordinate_system(query::barycentric);
        if (argvars.count("jacobi")) query::set_coordinate_system(query::jacobi);
        std::cout.flush();
        if (argvars.count("keplerian")) {
            query::set_keplerian_output();
            std::cerr << "# Output in Keplerian coordinates  \n";
        } else
            std::cerr << "# Output in Cartesian coordinates  \n";
        query::execute(argvars["logfile"], T, sys, body_range);
    } else if (command == "generate") {
        INFO_OUTPUT(1, "Generating new ensemble:  " << cfg["nsys"] << ", " << cfg["nbod"] << endl);
        current_ens = generate_ensemble(cfg);
        save_ensemble();
    } else if (command == "convert") {
        if (read_input_file(current_ens) && save_ensemble()) {
            INFO_OUTPUT(1, "Converted!");
        } else {
            INFO_OUTPUT(1, "Convertion failed");
            exit(1);
        }
    } else
        std::cerr << "Valid commands are: integrate, benchmark, verify, test, query, generate " << std::endl;
    return 0;
}

int main(int argc, char *argv[]) {
    try {
        return real_main(argc, argv);
    } catch (std::exception &e) {
        std::cerr << "terminate called after throwing an instance of '" << "swarm::runtime_error" << "'\n  what():  " << e.what()
                  << std::endl;
        return 1;
    }
}
