// config_reader.hpp -- mirror of ReadBouncer's ConfigReader for the classify/build usages
// (src/config/configReader.hpp:40-104, src/config/configReader.cpp:34-96, 232-345).  Same TOML keys,
// defaults and error messages; [MinKNOW] and [Basecaller] are parsed for completeness only.
#pragma once
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>

#include "toml_lite.hpp"

class ConfigReaderException : public std::exception
{
    std::string msg_;
public:
    explicit ConfigReaderException(const std::string& m) : msg_(m) {}
    const char* what() const noexcept override { return msg_.c_str(); }
};

class ConfigReader
{
public:
    toml_lite::Document configuration_{};
    std::filesystem::path output_dir{};
    std::filesystem::path log_dir{};
    std::string usage;

    struct IBF_Params  // configReader.hpp:55-66 (member defaults) and configReader.cpp:238-243 (parse defaults)
    {
        int size_k = 13;
        int fragment_size = 100000;
        int threads = 1;
        std::vector<std::filesystem::path> target_files{};
        std::vector<std::filesystem::path> deplete_files{};
        std::vector<std::filesystem::path> read_files{};
        double error_rate = 0.1;
        int chunk_length = 360;
        int max_chunks = 1;
    } IBF_Parsed;

    struct MinKNOW_Params
    {
        std::string host = "127.0.0.1";
        std::string port = "9501";
        std::string flowcell{};
        uint16_t minChannel = 1;
        uint16_t maxChannel = 512;
        std::filesystem::path token_path{};
    } MinKNOW_Parsed;

    struct Basecaller_Params
    {
        std::string caller = "DeepNano";
        std::string guppy_host = "127.0.0.1";
        std::string guppy_port = "5555";
        int basecall_threads = 3;
        std::string guppy_config = "dna_r9.4.1_450bps_fast";
    } Basecaller_Parsed;

    ConfigReader() = default;
    explicit ConfigReader(std::string const tomlFile) : tomlInputFile(tomlFile)
    {
        try {
            configuration_ = toml_lite::Document::parse_file(tomlFile);
        } catch (const std::exception& e) {
            throw ConfigReaderException(e.what());
        }
    }

    // configReader.cpp:59-91: the three top-level keys are mandatory; directories are created
    void parse_general()
    {
        try {
            log_dir = std::filesystem::path(configuration_.get_string("", "log_directory")).make_preferred();
            if (!std::filesystem::is_directory(log_dir) || !std::filesystem::exists(log_dir))
                std::filesystem::create_directories(log_dir);
            output_dir = std::filesystem::path(configuration_.get_string("", "output_directory")).make_preferred();
            if (!std::filesystem::is_directory(output_dir) || !std::filesystem::exists(output_dir))
                std::filesystem::create_directories(output_dir);
            usage = configuration_.get_string("", "usage");
        } catch (const std::exception& e) {
            throw ConfigReaderException(e.what());
        }
    }

    // ConfigReader::filterException, configReader.cpp:210-224: is this file an IBF?
    bool filterException(std::filesystem::path& file);

    void parse()  // configReader.cpp:427-438
    {
        readIBF();
        if (usage == "target") {
            readMinKNOW();
            readBasecaller();
        }
    }

    // ConfigReader::createLog, configReader.cpp:98-200: appends a [usage] table with the effective settings to
    // <output_directory>/configLog.toml (same key names, incl. the dashes of kmer-size / fragment-size)
    void createLog(std::string& usage_)
    {
        std::filesystem::path configLog(output_dir);
        configLog /= "configLog.toml";
        std::ofstream out(configLog, std::ios::app | std::ios::out);
        auto arr = [](const std::vector<std::filesystem::path>& v) {
            std::string s = "[";
            for (size_t i = 0; i < v.size(); ++i) s += std::string(i ? "," : "") + "\"" + v[i].string() + "\"";
            return s + "]";
        };
        out << "[" << usage_ << "]\n";
        out << "target_files = " << arr(IBF_Parsed.target_files) << "\n";
        out << "deplete_files = " << arr(IBF_Parsed.deplete_files) << "\n";
        if (usage_ != "build") out << "read_files = " << arr(IBF_Parsed.read_files) << "\n";
        out << "kmer-size = " << IBF_Parsed.size_k << "\n";
        out << "threads = " << IBF_Parsed.threads << "\n";
        out << "fragment-size = " << IBF_Parsed.fragment_size << "\n";
        if (usage_ != "build") {
            out << "exp_seq_error_rate = " << IBF_Parsed.error_rate << "\n";
            out << "chunk_length = " << IBF_Parsed.chunk_length << "\n";
            out << "max_chunks = " << IBF_Parsed.max_chunks << "\n";
        }
        out << "\n";
    }

    // echo of the effective configuration for --dump-config
    std::string dump() const
    {
        auto list = [](const std::vector<std::filesystem::path>& v) {
            std::string s = "[";
            for (size_t i = 0; i < v.size(); ++i) s += std::string(i ? ", " : "") + "'" + v[i].string() + "'";
            return s + "]";
        };
        std::string o;
        o += "usage              = \"" + usage + "\"\n";
        o += "output_directory   = '" + output_dir.string() + "'\n";
        o += "log_directory      = '" + log_dir.string() + "'\n\n[IBF]\n";
        o += "kmer_size          = " + std::to_string(IBF_Parsed.size_k) + "\n";
        o += "fragment_size      = " + std::to_string(IBF_Parsed.fragment_size) + "\n";
        o += "threads            = " + std::to_string(IBF_Parsed.threads) + "\n";
        o += "target_files       = " + list(IBF_Parsed.target_files) + "\n";
        o += "deplete_files      = " + list(IBF_Parsed.deplete_files) + "\n";
        o += "read_files         = " + list(IBF_Parsed.read_files) + "\n";
        char buf[64];
        snprintf(buf, sizeof buf, "%.17g", IBF_Parsed.error_rate);
        o += std::string("exp_seq_error_rate = ") + buf + "\n";
        o += "chunk_length       = " + std::to_string(IBF_Parsed.chunk_length) + "\n";
        o += "max_chunks         = " + std::to_string(IBF_Parsed.max_chunks) + "\n";
        return o;
    }

private:
    std::string tomlInputFile{};

    void readIBF()  // configReader.cpp:232-345
    {
        try {
            IBF_Parsed.size_k = (int)configuration_.get_int_or("IBF", "kmer_size", 13);
            IBF_Parsed.fragment_size = (int)configuration_.get_int_or("IBF", "fragment_size", 100000);
            IBF_Parsed.threads = (int)configuration_.get_int_or("IBF", "threads", 1);
            IBF_Parsed.error_rate = configuration_.get_double_or("IBF", "exp_seq_error_rate", 0.1);
            IBF_Parsed.chunk_length = (int)configuration_.get_int_or("IBF", "chunk_length", 250);
            IBF_Parsed.max_chunks = (int)configuration_.get_int_or("IBF", "max_chunks", 5);
        } catch (const std::exception& e) {
            throw ConfigReaderException(e.what());
        }
        auto read_list = [&](const char* key, std::vector<std::filesystem::path>& dst) {
            if (!configuration_.has("IBF", key)) return;  // "sometimes we only want to specify deplete files"
            try {
                for (const std::string& s : configuration_.get_string_array("IBF", key))
                    dst.emplace_back(std::filesystem::path(s).make_preferred());
            } catch (const std::exception& e) {
                throw ConfigReaderException(e.what());
            }
        };
        read_list("target_files", IBF_Parsed.target_files);
        read_list("deplete_files", IBF_Parsed.deplete_files);
        if (usage != "test" && IBF_Parsed.deplete_files.size() + IBF_Parsed.target_files.size() == 0)
            throw ConfigReaderException("[Error] At least one target or deplete file has to be specified!");
        for (const auto& file : IBF_Parsed.target_files)
            if (!std::filesystem::exists(file))
                throw ConfigReaderException("[Error] The following target file does not exist: " + file.string());
        for (const auto& file : IBF_Parsed.deplete_files)
            if (!std::filesystem::exists(file))
                throw ConfigReaderException("[Error] The following deplete file does not exist: " + file.string());
        std::vector<std::string> rf_tmp;
        if (configuration_.has("IBF", "read_files")) {
            try {
                rf_tmp = configuration_.get_string_array("IBF", "read_files");
            } catch (const std::exception& e) {
                throw ConfigReaderException(e.what());
            }
        } else if (usage == "classify") {
            throw ConfigReaderException("toml: key 'read_files' not found in [IBF]");
        }
        for (const std::string& file : rf_tmp) {
            std::filesystem::path rf = std::filesystem::path(file).make_preferred();
            if (!std::filesystem::exists(rf))
                throw ConfigReaderException("[Error] The following read file does not exist: " + rf.string());
            IBF_Parsed.read_files.emplace_back(std::move(rf));
        }
    }

    void readMinKNOW()  // configReader.cpp:353-385 (keys only; the MinKNOW client is out of scope)
    {
        MinKNOW_Parsed.host = configuration_.get_string_or("MinKNOW", "host", "127.0.0.1");
        MinKNOW_Parsed.port = configuration_.get_string_or("MinKNOW", "port", "9501");
        MinKNOW_Parsed.flowcell = configuration_.get_string_or("MinKNOW", "flowcell", "");
        MinKNOW_Parsed.token_path = configuration_.get_string_or("MinKNOW", "token_path", "");
        if (configuration_.has("MinKNOW", "channels")) {
            auto ch = configuration_.get_int_array("MinKNOW", "channels");
            if (ch.size() == 2) {
                MinKNOW_Parsed.minChannel = (uint16_t)ch[0];
                MinKNOW_Parsed.maxChannel = (uint16_t)ch[1];
            }
        }
    }

    void readBasecaller()  // configReader.cpp:393-420
    {
        Basecaller_Parsed.caller = configuration_.get_string_or("Basecaller", "caller", "DeepNano");
        Basecaller_Parsed.guppy_host = configuration_.get_string_or("Basecaller", "host", "127.0.0.1");
        Basecaller_Parsed.guppy_port = configuration_.get_string_or("Basecaller", "port", "5555");
        Basecaller_Parsed.basecall_threads = (int)configuration_.get_int_or("Basecaller", "threads", 3);
        Basecaller_Parsed.guppy_config = configuration_.get_string_or("Basecaller", "config", "dna_r9.4.1_450bps_fast");
    }
};
