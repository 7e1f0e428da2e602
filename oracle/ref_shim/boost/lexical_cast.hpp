// TEST INFRASTRUCTURE ONLY: boost::lexical_cast for the CIGAR parser in basics/cigar_string.cpp (unused by the bridge).
#pragma once
#include <sstream>
#include <string>
namespace boost { template <class T, class S> T lexical_cast(const S& s) { std::istringstream is {std::string {s}}; T v {}; is >> v; return v; } }
