// seqio.hpp -- FASTA/FASTQ record reader and FASTA writer for the host driver (the reference uses
// seqan::SeqFileIn/SeqFileOut, src/main/classify.hpp:217-237,301).  Format is detected per record from the
// first character ('>' FASTA, '@' FASTQ); multi-line FASTA, CRLF and blank lines are accepted.
#pragma once
#include <fstream>
#include <stdexcept>
#include <string>

namespace seqio
{

class Reader
{
    std::ifstream in_;
    std::string pending_;  // a header line already consumed while reading the previous FASTA record
    bool have_pending_ = false;

    static void chomp(std::string& s)
    {
        while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back();
    }
    bool next_line(std::string& line)
    {
        if (have_pending_) {
            line = pending_;
            have_pending_ = false;
            return true;
        }
        if (!std::getline(in_, line)) return false;
        chomp(line);
        return true;
    }

public:
    explicit Reader(const std::string& path) : in_(path, std::ios_base::binary) {}
    bool is_open() const { return in_.is_open(); }

    // returns false at end of file; throws on malformed input
    bool read_record(std::string& id, std::string& seq)
    {
        std::string line;
        do {
            if (!next_line(line)) return false;
        } while (line.empty());
        id.clear();
        seq.clear();
        if (line[0] == '>') {
            id = line.substr(1);
            while (next_line(line)) {
                if (!line.empty() && line[0] == '>') {
                    pending_ = line;
                    have_pending_ = true;
                    break;
                }
                seq += line;
            }
            return true;
        }
        if (line[0] == '@') {
            id = line.substr(1);
            std::string plus, qual;
            if (!next_line(seq)) throw std::runtime_error("FASTQ: truncated record " + id);
            // multi-line FASTQ is not produced by basecallers; a single sequence line is assumed
            if (!next_line(plus) || plus.empty() || plus[0] != '+') throw std::runtime_error("FASTQ: '+' line expected in " + id);
            if (!next_line(qual)) throw std::runtime_error("FASTQ: quality line missing in " + id);
            return true;
        }
        throw std::runtime_error("unrecognised sequence record starting with '" + line.substr(0, 10) + "'");
    }
};

inline void write_fasta(std::ostream& out, const std::string& id, const std::string& seq)
{
    out << ">" << id << "\n" << seq << "\n";
}

}  // namespace seqio
